"""Drop-in alias: ``from ssd import LLM, SamplingParams`` (reference ssd/__init__.py:1-2)."""
from ssd_amd.sampling_params import SamplingParams  # noqa: F401


def __getattr__(name):
    if name == "LLM":
        from ssd_amd.llm import LLM
        return LLM
    raise AttributeError(name)
