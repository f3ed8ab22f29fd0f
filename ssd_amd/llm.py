from ssd_amd.engine.llm_engine import LLMEngine


class LLM(LLMEngine):
    """Drop-in for ``ssd.LLM`` (reference ssd/llm.py:1-5)."""
