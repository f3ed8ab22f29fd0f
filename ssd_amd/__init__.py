"""ssd_amd: MI355X-native speculative-decoding engine with the API of tanishqkumar/ssd
(``from ssd_amd import LLM, SamplingParams``; the ``ssd`` alias package re-exports the same names)."""
from ssd_amd.sampling_params import SamplingParams  # noqa: F401


def __getattr__(name):          # lazy: importing the package must not require a GPU / the HIP library
    if name == "LLM":
        from ssd_amd.llm import LLM
        return LLM
    raise AttributeError(name)
