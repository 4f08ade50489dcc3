"""Another build of libcrispy_hip.so as the library `crispy_amd` calls, for the span of a `with` block.

`libcrispy_hip_dev.so` (`make dev`, part of `build()`) is the product's sources with the developer knobs compiled in
(crispy_amd/csrc/api_util.h: dev_env).  One of them, CRISPY_ASR_DECODE=stages, runs every decode step as one launch per
stage: the second implementation of the decoder's arithmetic that the fused step kernels are compared with.  The two forms
are not bit-identical, so the release library does not read the variable (a host's environment must not change a
transcript); tests that want the staged form build their engine inside `staged_decoder()`.

Objects created inside the block belong to the variant library: use and close them inside it."""
import contextlib
import os


@contextlib.contextmanager
def library_variant(name, env=None):
    from crispy_amd import _native as N
    N.lib()                                   # the release library first: its handle is what the block restores
    old_lib = N._lib
    old_env = {k: os.environ.get(k) for k in (env or {})}
    N._lib = N.load_variant(name)
    os.environ.update(env or {})
    try:
        yield N._lib
    finally:
        N._lib = old_lib
        for k, v in old_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def staged_decoder():
    """The developer build with every decode step as one launch per stage."""
    return library_variant("dev", {"CRISPY_ASR_DECODE": "stages"})
