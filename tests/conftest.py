import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle's bf16 GEMV-shaped matmuls stop scaling -- and then collapse -- beyond a few tens of threads: on the GPU box's
    # 256-core host the full-size oracle engines (8B target + 1B draft) ran minutes per test with torch's default of one thread per
    # core (bench.py's cpu_baseline leg found the same: 256 threads ~1000x slower than 16).  Results do not depend on the count.
    import torch
    torch.set_num_threads(min(16, os.cpu_count() or 1))


# Tests that start SEVERAL processes on the one test GPU (ranks that spin on each other's flags in the one-shot all-reduce,
# self-launched bench runs) go last: every single-process parity test has reported before them.
_MULTI_PROCESS_GPU = ("test_bench_gpu.py", "test_custom_ar_gpu.py", "test_tp_one_gpu.py")


def pytest_collection_modifyitems(config, items):
    # SSD_TEST_SHUFFLE_SEED=n: run the test FILES in a seeded random order (tests inside a file keep theirs) -- the round-end
    # validation runs the GPU suite three times in fresh processes with three orders (profiles/r06_gpu_suite_shuffle*.txt), so a
    # result that depends on what ran before it (allocator state, leftover stream work, lazily loaded code objects) shows up.
    seed = os.environ.get("SSD_TEST_SHUFFLE_SEED")
    if seed:
        import random
        files = []
        for it in items:
            f = it.nodeid.split("::")[0]
            if f not in files:
                files.append(f)
        random.Random(int(seed)).shuffle(files)
        rank = {f: i for i, f in enumerate(files)}
        items.sort(key=lambda it: rank[it.nodeid.split("::")[0]])       # (stable: keeps the order inside a file)
        print(f"\n[conftest] file order (seed {seed}): {' '.join(os.path.basename(f) for f in files)}")
    last = [it for it in items if any(name in it.nodeid for name in _MULTI_PROCESS_GPU)]
    if last:
        rest = [it for it in items if it not in last]
        items[:] = rest + last


@pytest.fixture(scope="session")
def golden():
    from oracle.io import load_npz

    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_npz(os.path.join(GOLDEN, name + ".npz"))
        return cache[name]

    return get
