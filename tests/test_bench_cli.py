"""bench/bench.py: the reference's command line (bench/bench.py:15-89 of tanishqkumar/ssd) on this engine."""
import importlib.util
import os
import subprocess
import sys
from types import SimpleNamespace

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_cli():
    spec = importlib.util.spec_from_file_location("ssd_bench_cli", os.path.join(ROOT, "bench", "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_model_and_prompt_resolution(tmp_path, monkeypatch):
    cli = load_cli()
    monkeypatch.delenv("SSD_HF_CACHE", raising=False)
    a = SimpleNamespace(llama=True, size="70", draft=None, eagle=False)
    assert cli.resolve_models(a) == ("Llama-3.1-70B-Instruct", "llama-3.1-70b", "llama-3.2-1b")
    # --eagle picks the model's EAGLE-3 draft (bench_helpers.py:50-63)
    assert cli.resolve_models(SimpleNamespace(llama=True, size="8", draft=None, eagle=True))[2] == "eagle3-llama-3.1-8b"
    assert cli.resolve_models(SimpleNamespace(llama=True, size="70", draft=None, eagle=True))[2] == "eagle3-llama-3.3-70b"
    q = SimpleNamespace(llama=False, size="32", draft="0.6", eagle=False)
    assert cli.resolve_models(q) == ("Qwen3-32B", "qwen3-32b", "qwen3-0.6b")
    # a HF-cache style tree is picked up (snapshots/<hash>/config.json), as in bench_helpers.py:14-43
    snap = tmp_path / "models--meta-llama--Llama-3.2-1B-Instruct" / "snapshots" / "abc"
    snap.mkdir(parents=True)
    (snap / "config.json").write_text("{}")
    monkeypatch.setenv("SSD_HF_CACHE", str(tmp_path))
    name, model, draft = cli.resolve_models(SimpleNamespace(llama=True, size="1", draft=None, eagle=False))
    assert model == str(snap) and draft == str(snap)
    args = SimpleNamespace(random=True, input_len=12, numseqs=3)
    prompts, tok = cli.load_prompts(args, "llama-3.2-1b")
    assert tok is None and len(prompts) == 3 and all(len(p) == 12 and all(0 <= t <= 10000 for t in p) for p in prompts)
    # every --size the reference offers has a preset with shapes the kernels accept
    from ssd_amd.model_config import PRESETS
    for table in (cli.LLAMA, cli.QWEN):
        for _, preset in table.values():
            c = PRESETS[preset]
            assert c.hidden_size % 32 == 0 and c.intermediate_size % 32 == 0 and c.num_heads % c.num_kv_heads == 0
            assert c.head_dim in (64, 128) and ((c.num_heads + 2 * c.num_kv_heads) * c.head_dim) % 16 == 0


@pytest.mark.gpu
def test_reference_command_line_runs_on_the_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = dict(os.environ, PYTHONPATH=ROOT)
    env.pop("SSD_HF_CACHE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench", "bench.py"), "--size", "1", "--draft", "1", "--spec", "--k", "3", "--random",
           "--numseqs", "2", "--b", "2", "--output_len", "24", "--input_len", "32", "--max_model_len", "1024"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    assert "Total Throughput:" in p.stdout and "Avg tokens per step" in p.stdout and "Engine exited!" in p.stdout
