from .lib import load_library, lib_path, build_library  # noqa: F401
