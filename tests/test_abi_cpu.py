"""CPU-only checks of the C-ABI boundary: libssv_hip.so loads without a GPU, exports every symbol that
include/ssv_hip.h declares, answers its host-side queries, and rejects bad arguments with the documented
codes before touching the device."""
import ctypes
import os
import re

import pytest

from spoofsv_amd import _lib


def test_header_declares_expected_entry_points():
    protos = _lib.parse_header()
    must = ["ssv_version", "ssv_arch", "ssv_last_error", "ssv_set_precision", "ssv_conv1d_fwd", "ssv_conv1d_bwd_data", "ssv_conv1d_bwd_weight",
            "ssv_channel_ln_act_fwd", "ssv_channel_ln_act_bwd", "ssv_highway_conv1d_fwd", "ssv_highway_conv1d_bwd",
            "ssv_text_embed_fwd", "ssv_text_embed_bwd", "ssv_attention_train_fwd", "ssv_attention_train_bwd",
            "ssv_attention_step", "ssv_attention_apply", "ssv_deconv1d_k2s2_fwd", "ssv_deconv1d_k2s2_bwd",
            "ssv_spec_losses_fwd", "ssv_spec_losses_bwd", "ssv_guided_att_loss_fwd", "ssv_guided_att_loss_bwd",
            "ssv_adam_multi", "ssv_lstm_fwd", "ssv_proj_l2norm_fwd", "ssv_ge2e_loss_fwd",
            "ssv_spec_losses_fwd_bwd", "ssv_deinterleave2_rows_amax", "ssv_bias_grad"]          # ABI 7
    for name in must:
        assert name in protos, name
    # every prototype in the header text was parsed (count `ssv_xxx(` occurrences outside comments)
    src = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER).read(), flags=re.S)
    names = set(re.findall(r"\b(ssv_\w+)\s*\(", src))
    assert names == set(protos), names ^ set(protos)


def test_library_exports_every_declared_symbol():
    L = _lib.lib()                      # raises if a declared symbol is missing
    assert L.ssv_version() == 7
    assert L.ssv_arch() == b"gfx950"
    # the default arithmetic is split-fp16 (2) unless SSV_PRECISION names another mode (strictly parsed: a typo is an error, see the test below)
    default = {"fp32": 0, "0": 0, "bf16x3": 1, "1": 1}.get(os.environ.get("SSV_PRECISION", ""), 2)
    prev = L.ssv_set_precision(0)
    assert prev == default and L.ssv_get_precision() == 0
    assert L.ssv_set_precision(prev) == 0 and L.ssv_get_precision() == default
    assert L.ssv_amax_rows(325) == 24 and L.ssv_amax_rows(1300) == 84
    raw = ctypes.CDLL(_lib.LIBPATH)
    for name in _lib.parse_header():
        assert hasattr(raw, name), name


def test_unknown_precision_value_is_an_error_of_the_query_entries_not_an_abort():
    """A C host (no Python loader in front) with SSV_PRECISION misspelt: ssv_get_precision() returns SSV_UNSUPPORTED with a message, and an
    explicit ssv_set_precision(mode) overrides the variable -- the process lives.  (spoofsv_amd._lib raises on the same value before the first call.)"""
    import subprocess, sys
    code = ("import ctypes, sys; L = ctypes.CDLL(sys.argv[1]); L.ssv_last_error.restype = ctypes.c_char_p\n"
            "r = L.ssv_get_precision(); msg = L.ssv_last_error()\n"
            "assert r == -2 and b'fp33' in msg, (r, msg)\n"
            "assert L.ssv_set_precision(0) == -2 and L.ssv_get_precision() == 0\n"
            "assert L.ssv_set_precision(2) == 0 and L.ssv_get_precision() == 2\n"
            "print('alive')")
    env = dict(os.environ, SSV_PRECISION="fp33")
    r = subprocess.run([sys.executable, "-c", code, _lib.LIBPATH], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "alive" in r.stdout, (r.returncode, r.stdout, r.stderr)
    root = os.path.dirname(os.path.dirname(os.path.abspath(_lib.HEADER)))
    r = subprocess.run([sys.executable, "-c", "from spoofsv_amd import _lib; _lib.lib()"], env=dict(env, PYTHONPATH=root), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "SSV_PRECISION" in r.stderr, (r.returncode, r.stderr[-300:])        # the Python loader raises before the first call


def test_no_torch_types_in_signatures():
    src = open(_lib.HEADER).read()
    assert "torch" not in src.lower().replace("pytorch", "").replace("torch's", "").replace("torch layout", "").replace("on torch", "").replace("torch.cat", "")
    assert "at::" not in src and "Tensor" not in src


def test_workspace_queries_are_host_only():
    q = _lib.query
    B, C, L, k = 32, 256, 325, 3
    assert q("ssv_conv1d_bwd_data_workspace", C, 2 * C, k) >= 4 * C * 2 * C * k
    assert q("ssv_highway_conv1d_bwd_workspace", B, C, L, k) >= 4 * B * 2 * C * L
    assert q("ssv_attention_train_bwd_workspace", B, 256, 186, 325) >= 4 * B * 186 * 325
    assert q("ssv_lstm_fwd_workspace", 880, 120, 40, 768, 3) >= 4 * 120 * 4 * 768 * 880
    assert q("ssv_ge2e_loss_fwd_workspace", 88, 10, 256) >= 4 * 88 * 256


def test_bad_arguments_fail_before_the_device():
    L = _lib.lib()
    null = ctypes.c_void_p(0)
    one = ctypes.c_void_p(16)           # non-null dummy, never dereferenced: the checks come first
    # empty problem -> -1
    rc = L.ssv_conv1d_fwd(one, 0, null, 0, one, null, null, null, one, 0, null, 0, 4, 4, 8, 3, 1, 0, null, 0, null)
    assert rc == -1 and b"conv1d_fwd" in L.ssv_last_error()
    # unsupported kernel size -> -2
    rc = L.ssv_conv1d_fwd(one, 64, null, 0, one, null, null, null, one, 64, null, 1, 4, 4, 16, 5, 1, 0, null, 0, null)
    assert rc == -2 and b"kernel_size" in L.ssv_last_error()
    # dilation halo beyond the staged tile -> -2
    rc = L.ssv_conv1d_fwd(one, 64, null, 0, one, null, null, null, one, 64, null, 1, 4, 4, 16, 3, 28, 0, null, 0, null)
    assert rc == -2
    # workspace too small -> -1
    rc = L.ssv_conv1d_bwd_data(one, 64, null, 0, one, null, null, one, 64, 1, 4, 4, 16, 3, 1, 0, one, 8, null)
    assert rc == -1 and b"workspace" in L.ssv_last_error()
    with pytest.raises(RuntimeError):
        _lib.call("ssv_ge2e_loss_fwd", one, one, one, one, null, 4, 1, 8, one, 1 << 20, null)   # M must be > 1
    # the later additions check their arguments on the host as well (vocoder, second order, column-incremental synthesis)
    rc = L.ssv_ola_frames(one, one, one, 1, 1024, 2, 256, null)                                 # hop*(T-1) <= N/2
    assert rc == -1 and b"ola_frames" in L.ssv_last_error()
    rc = L.ssv_frame_signal(one, one, 1, 100, 1024, 1, 256, null)                               # n <= N/2: no reflect padding
    assert rc == -1
    rc = L.ssv_channel_ln_bwd2(one, 0, one, 0, one, 0, one, one, one, 0, one, 0, one, 1, 8, 4, one, 0, null)
    assert rc == -1 and b"workspace" in L.ssv_last_error()
    rc = L.ssv_column_matvec(one, null, null, 0, one, 64, null, 0, 0, null, 1, one, 64, 2, 62, 8, 1, null)   # C not a multiple of 4
    assert rc == -2 and b"multiple of 4" in L.ssv_last_error()
    rc = L.ssv_column_matvec(one, null, null, 0, one, 64, null, 0, 0, null, 1, one, 64, 2, 64, 8, 3, null)   # k = 3 without a history
    assert rc == -1 and b"history" in L.ssv_last_error()
    rc = L.ssv_column_ln_act(one, 0, one, one, one, 0, 2, 5000, 0, null)
    assert rc == -2


def test_ops_fail_loudly_on_cpu_tensors():
    import torch
    from spoofsv_amd.tts import SSRN
    m = SSRN(80, 65, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.rand(1, 80, 8))


def test_conv_pack_plan_is_host_only_and_consistent():
    """ssv_conv_pack_plan runs on the host: job table for two weights (forward + transposed order each)."""
    import ctypes
    from spoofsv_amd import _lib
    L = _lib.lib()
    n = 2
    vp = ctypes.c_void_p
    w = (vp * n)(0x1000, 0x2000)
    pl = (vp * n)(0x100000, 0x200000)
    co = (ctypes.c_int * n)(512, 80)
    ci = (ctypes.c_int * n)(256, 513)
    kk = (ctypes.c_int * n)(3, 1)
    jobs = (_lib.PackJob * (2 * n))()
    nblocks = L.ssv_conv_pack_plan(n, w, pl, co, ci, kk, jobs)
    assert nblocks > 0
    # forward job of weight 0: rows = Cout, K = Cin, element (m, c, tap) at w[m*Cin*k + c*k + tap]
    j = jobs[0]
    assert (j.M, j.K, j.Kpad, j.KT, j.sm, j.sk, j.first_block) == (512, 256, 256, 3, 768, 3, 0)
    # transposed job: rows = Cin, K = Cout; its planes follow the forward planes inside the same buffer
    t = jobs[1]
    assert (t.M, t.K, t.KT, t.sm, t.sk) == (256, 512, 3, 3, 768)
    fwd_bytes = 2 * 3 * 512 * 256 * 2
    assert t.planes - j.planes == fwd_bytes
    assert L.ssv_conv_pack_bytes(512, 256, 3) == 2 * fwd_bytes + 256       # + the two inverse scales of the split-fp16 planes
    assert j.inv_out - j.planes == 2 * fwd_bytes and t.inv_out - j.inv_out == 128
    assert L.ssv_conv_pack_multi_workspace(2 * n) >= 4 * 32 * n
    # ragged weight: K padded to 32, rows to 16
    r = jobs[2]
    assert (r.M, r.K, r.Kpad, r.KT) == (80, 513, 544, 1)
    firsts = [jobs[i].first_block for i in range(2 * n)]
    assert firsts == sorted(firsts) and firsts[0] == 0 and firsts[-1] < nblocks
    assert L.ssv_conv_pack_plan(1, w, pl, co, ci, (ctypes.c_int * 1)(2), jobs) < 0      # kernel size 2 is not a conv weight here
