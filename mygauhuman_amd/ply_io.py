"""PLY input / output with the on-disk formats of the reference, written against numpy only (the
reference uses the `plyfile` package, which is not a dependency here):

    fetch_ply / store_ply             scene/dataset_readers.py:130-153  (x y z nx ny nz float32, red green blue uint8)
    save_gaussians_ply / load_...     scene/gaussian_model.py:309-346, 353-407: one float32 property per column, in the order
                                      x y z nx ny nz ar ag ab roughness f_dc_* f_rest_* opacity scale_* rot_*
                                      (SH stored channel-major: features.transpose(1, 2).flatten)
Files are `binary_little_endian 1.0` with a single `vertex` element and the minimal header plyfile writes; the reader also
accepts ascii files and the other scalar property types.  Pinned by the reference's own data file check/points3d.ply
(its first 256 vertices are tests/golden/points3d_head256.ply): read + write reproduces them byte for byte.
"""
import os

import numpy as np

_TYPES = {"char": "i1", "uchar": "u1", "short": "i2", "ushort": "u2", "int": "i4", "uint": "u4", "float": "f4", "double": "f8",
          "int8": "i1", "uint8": "u1", "int16": "i2", "uint16": "u2", "int32": "i4", "uint32": "u4", "float32": "f4", "float64": "f8"}
_NAMES = {"i1": "char", "u1": "uchar", "i2": "short", "u2": "ushort", "i4": "int", "u4": "uint", "f4": "float", "f8": "double"}


def read_ply(path):
    """Returns a numpy structured array with one field per property of the `vertex` element."""
    with open(path, "rb") as f:
        data = f.read()
    end = data.find(b"end_header")
    if not data.startswith(b"ply") or end < 0:
        raise ValueError(f"{path}: not a PLY file")
    header_end = data.find(b"\n", end) + 1
    fmt, count, props, in_vertex = None, 0, [], False
    for line in data[:end].decode("ascii").splitlines():
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "format":
            fmt = tok[1]
        elif tok[0] == "element":
            in_vertex = tok[1] == "vertex"
            if in_vertex:
                count = int(tok[2])
        elif tok[0] == "property" and in_vertex:
            if tok[1] == "list":
                raise ValueError(f"{path}: list properties on the vertex element are not supported")
            props.append((tok[2], _TYPES[tok[1]]))
    if fmt == "binary_little_endian":
        dt = np.dtype([(n, "<" + t) for n, t in props])
        return np.frombuffer(data, dtype=dt, count=count, offset=header_end).copy()
    if fmt == "binary_big_endian":
        dt = np.dtype([(n, ">" + t) for n, t in props])
        return np.frombuffer(data, dtype=dt, count=count, offset=header_end).astype(np.dtype([(n, "<" + t) for n, t in props]))
    if fmt == "ascii":
        rows = data[header_end:].decode("ascii").split("\n")[:count]
        out = np.empty(count, dtype=np.dtype([(n, "<" + t) for n, t in props]))
        for i, row in enumerate(rows):
            vals = row.split()
            for (n, t), v in zip(props, vals):
                out[n][i] = float(v) if t[0] == "f" else int(v)
        return out
    raise ValueError(f"{path}: unknown PLY format {fmt}")


def write_ply(path, elements):
    """elements: structured array (little-endian scalar fields) -> binary_little_endian PLY with one `vertex` element."""
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    lines = ["ply", "format binary_little_endian 1.0", f"element vertex {len(elements)}"]
    for name in elements.dtype.names:
        lines.append(f"property {_NAMES[elements.dtype[name].str[1:]]} {name}")
    lines.append("end_header")
    with open(path, "wb") as f:
        f.write(("\n".join(lines) + "\n").encode("ascii"))
        f.write(np.ascontiguousarray(elements).tobytes())


def fetch_ply(path):
    """scene/dataset_readers.py:130-136 -> (points [N,3] float32, colors [N,3] in [0,1] float64, normals [N,3] float32)."""
    v = read_ply(path)
    positions = np.vstack([v["x"], v["y"], v["z"]]).T
    colors = np.vstack([v["red"], v["green"], v["blue"]]).T / 255.0
    normals = np.vstack([v["nx"], v["ny"], v["nz"]]).T
    return positions, colors, normals


def store_ply(path, xyz, rgb, normal):
    """scene/dataset_readers.py:138-153 (rgb 0..255)."""
    dtype = [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4"), ("red", "u1"), ("green", "u1"),
             ("blue", "u1")]
    el = np.empty(xyz.shape[0], dtype=dtype)
    for k, n in enumerate(("x", "y", "z")):
        el[n] = xyz[:, k]
    for k, n in enumerate(("nx", "ny", "nz")):
        el[n] = normal[:, k]
    for k, n in enumerate(("red", "green", "blue")):
        el[n] = rgb[:, k]
    write_ply(path, el)


def gaussian_attribute_names(n_dc, n_rest, n_scale=3, n_rot=4):
    """construct_list_of_attributes, scene/gaussian_model.py:309-326."""
    names = ["x", "y", "z", "nx", "ny", "nz", "ar", "ag", "ab", "roughness"]
    names += [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)] + ["opacity"]
    names += [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)]
    return names


def save_gaussians_ply(model, path):
    """GaussianModel.save_ply, scene/gaussian_model.py:329-346 (raw, un-activated parameters)."""
    t = lambda x: x.detach().cpu().numpy()  # noqa: E731
    f_dc = model._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    f_rest = model._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    cols = np.concatenate((t(model._xyz), t(model._normal), t(model._albedo), t(model._roughness).reshape(len(f_dc), -1)[:, :1], f_dc,
                           f_rest, t(model._opacity), t(model._scaling), t(model._rotation)), axis=1).astype(np.float32)
    names = gaussian_attribute_names(f_dc.shape[1], f_rest.shape[1], model._scaling.shape[1], model._rotation.shape[1])
    assert cols.shape[1] == len(names)
    el = np.empty(cols.shape[0], dtype=[(n, "<f4") for n in names])
    for k, n in enumerate(names):
        el[n] = cols[:, k]
    write_ply(path, el)


def load_gaussians_ply(model, path, device=None):
    """GaussianModel.load_ply, scene/gaussian_model.py:353-407: fills the raw parameters of `model` and sets
    active_sh_degree = max_sh_degree.  (The reference passes requires_grad_(False) for the opacity into nn.Parameter, whose
    own default turns it back on; same here.)"""
    import torch
    import torch.nn as nn
    v = read_ply(path)
    dev = device if device is not None else model.device
    names = v.dtype.names
    col = lambda n: np.asarray(v[n], np.float32)  # noqa: E731
    stack = lambda ns: np.stack([col(n) for n in ns], axis=1)  # noqa: E731
    by_index = lambda prefix: sorted([n for n in names if n.startswith(prefix)], key=lambda x: int(x.split("_")[-1]))  # noqa: E731
    xyz = stack(["x", "y", "z"])
    P = xyz.shape[0]
    features_dc = stack(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(P, 3, 1)
    extra = by_index("f_rest_")
    assert len(extra) == 3 * (model.max_sh_degree + 1) ** 2 - 3
    features_extra = stack(extra).reshape(P, 3, (model.max_sh_degree + 1) ** 2 - 1) if extra else np.zeros((P, 3, 0), np.float32)
    par = lambda a, grad=True: nn.Parameter(torch.tensor(a, dtype=torch.float, device=dev).requires_grad_(grad))  # noqa: E731
    model._xyz = par(xyz)
    model._features_dc = nn.Parameter(torch.tensor(features_dc, dtype=torch.float, device=dev).transpose(1, 2).contiguous().requires_grad_(True))
    model._features_rest = nn.Parameter(torch.tensor(features_extra, dtype=torch.float, device=dev).transpose(1, 2).contiguous().requires_grad_(True))
    model._opacity = par(col("opacity")[:, None], False)
    model._scaling = par(stack(by_index("scale_")))
    model._rotation = par(stack(by_index("rot")))
    model._normal = par(stack(["nx", "ny", "nz"]))
    model._albedo = par(stack(["ar", "ag", "ab"]))
    model._roughness = par(col("roughness")[:, None])
    model.active_sh_degree = model.max_sh_degree
    return model
