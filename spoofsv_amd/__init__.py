"""MI355X (gfx950) hot path of SpoofSV: hand-written HIP kernels behind a C ABI (libssv_hip.so) and the
host-side mirror of the reference's module interface.  See DESIGN.md / INTEGRATION.md."""

_MODES = {"fp32": 0, "bf16x3": 1, "f16x2": 2}
_NAMES = {v: k for k, v in _MODES.items()}


def set_precision(mode):
    """Arithmetic of the conv GEMMs (include/ssv_hip.h, ``ssv_set_precision``):

    * ``"f16x2"`` (default): fp32 operands scaled by a power of two and split into fp16 hi+lo (22 significand bits), three
      fp16 MFMAs per product, fp32 accumulate -- fp32-grade (~2^-22 per product) at the 16-bit MFMA rate;
    * ``"fp32"``: fp32-input MFMA, exact fp32 fma chains;
    * ``"bf16x3"``: bf16 hi+lo (~2^-16 per product), narrower than the reference's fp32 -- opt-in.

    Returns the previous mode.  Resident pre-split weight planes are written in the mode in force, so they are dropped here."""
    from . import _lib, resident
    prev = _lib.lib().ssv_set_precision(_MODES[mode])
    if prev != _MODES[mode]:
        resident.invalidate()
    return _NAMES[prev]


def get_precision():
    from . import _lib
    return _NAMES[_lib.precision()]
