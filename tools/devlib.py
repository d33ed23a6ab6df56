"""Developer builds of the operator library with extra -D flags (ST_DEV_CONFIGS, ST_PROBE, ...), kept OUT of the product
tree: tools/build_variant.sh <name> -D... writes tools/_variants/<name>/libstabletriton_amd.so and a tool selects it with

    from tools.devlib import use_variant; use_variant(os.environ.get("ST_VARIANT"))      # before any op runs

The product package never looks at the environment or at tools/_variants."""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def variant_path(name: str) -> str:
    return os.path.join(ROOT, "tools", "_variants", name, "libstabletriton_amd.so")


def use_variant(name=None):
    """Load tools/_variants/<name> as the operator library of this process (None / empty: the product build)."""
    from stabletriton_amd import _C
    return _C.load(variant_path(name) if name else None)
