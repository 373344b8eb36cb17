"""MI355X (gfx950) hot path of SpoofSV: hand-written HIP kernels behind a C ABI (libssv_hip.so) and the
host-side mirror of the reference's module interface.  See DESIGN.md / INTEGRATION.md."""


def set_precision(mode):
    """Arithmetic of the conv GEMMs: "bf16x3" (default; fp32 operands split into bf16 hi+lo, three bf16 MFMAs per
    product, fp32 accumulate, ~1e-5 relative) or "fp32" (fp32-input MFMA, exact fp32 fma chains).  Returns the
    previous mode."""
    from . import _lib
    modes = {"fp32": 0, "bf16x3": 1}
    prev = _lib.lib().ssv_set_precision(modes[mode])
    return "bf16x3" if prev else "fp32"
