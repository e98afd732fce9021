import os, sys
os.environ["GSR_FUZZ_SEEDS"] = "86"
import pytest
for i in range(12):
    rc = pytest.main(["-q", "-x", "tests/test_gpu_fuzz.py", "-k", "85]", "-p", "no:cacheprovider", "--tb=line"])
    print("run", i, "rc", rc, flush=True)
