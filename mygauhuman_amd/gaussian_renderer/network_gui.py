"""Stand-in for the reference's gaussian_renderer/network_gui.py (the SIBR viewer socket: out of scope) with
the module surface train.py touches (train.py:17,180-193,582): `conn` stays None, so the training loop's viewer block is skipped;
`init` / `try_connect` do nothing; `receive` / `send` raise, because without a connection train.py never reaches them."""
host = "127.0.0.1"
port = 6009
conn = None
addr = None


def init(wish_host, wish_port):
    global host, port
    host, port = wish_host, wish_port


def try_connect():
    return None


def receive():
    raise RuntimeError("network_gui: the viewer connection is not part of mygauhuman_amd (no connection is ever accepted)")


def send(message_bytes, verify):
    raise RuntimeError("network_gui: the viewer connection is not part of mygauhuman_amd (no connection is ever accepted)")
