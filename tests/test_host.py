"""CPU-only tests: the C-ABI library loads and exports every symbol include/sfa.h declares,
host-side logic of the op wrappers and of the HF/verl boundary.  No GPU compute."""
import ctypes
import os
import re

import pytest
import torch

import golden_util as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sfa.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sfa_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from sink_attention import _native
    lib = _native.lib()
    syms = _declared_symbols()
    assert {"sfa_fwd", "sfa_bwd", "sfa_decode", "sfa_bwd_workspace_bytes", "sfa_decode_workspace_bytes",
            "sfa_abi_version", "sfa_last_error", "sfa_last_path"} <= set(syms)
    for s in syms:
        assert hasattr(lib, s), s
    assert lib.sfa_abi_version() == _native.ABI_VERSION == 2


def test_descriptor_matches_header_layout():
    from sink_attention._native import SfaTensor, desc
    assert ctypes.sizeof(SfaTensor) == 8 + 32 + 32 + 8
    t = torch.zeros(2, 3, 5, 8).transpose(1, 2)
    d = desc(t)
    assert list(d.shape) == [2, 5, 3, 8] and list(d.stride) == [120, 8, 40, 1] and d.dtype == 0


def test_workspace_queries_need_no_gpu():
    from sink_attention import _native
    lib = _native.lib()
    # Delta [B,Hq,N] f32 is always part of the backward workspace
    assert lib.sfa_bwd_workspace_bytes(4, 32, 8, 8192, 128, 2, 4, 4096, 0) >= 4 * 32 * 8192 * 4
    assert lib.sfa_decode_workspace_bytes(32, 32, 32, 131072, 128, 2) > 0
    assert lib.sfa_decode_workspace_bytes(1, 4, 4, 100, 20, 2) == 0          # 40-byte rows: unsupported
    assert b"16 bytes" in lib.sfa_last_error()


@pytest.mark.parametrize("B,Hq,Hkv,D,cap", [(8, 32, 8, 128, 16388), (16, 16, 16, 64, 8196), (1, 8, 8, 128, 4100),
                                             (32, 32, 32, 128, 4100)])
def test_decode_workspace_is_monotonic_in_the_key_count(B, Hq, Hkv, D, cap):
    """A ring cache sizes its decode workspace once for num_sink + window_size keys and then steps through every
    fill level: the query must never report MORE for fewer keys (the planned split count alone is not monotonic)."""
    from sink_attention import _native
    lib = _native.lib()
    full = lib.sfa_decode_workspace_bytes(B, Hq, Hkv, cap, D, 2)
    prev = 0
    for n in list(range(1, 600)) + list(range(600, cap + 1, 37)) + [cap]:
        ws = lib.sfa_decode_workspace_bytes(B, Hq, Hkv, n, D, 2)
        assert 0 < ws <= full, (n, ws, full)
        assert ws >= prev, (n, ws, prev)
        prev = ws


def test_invalid_arguments_are_rejected_before_any_launch():
    from sink_attention import _native as N
    lib = N.lib()
    q = torch.zeros(1, 4, 8, 16)
    d = N.desc(q)
    d.ptr = None
    st = lib.sfa_fwd(d, d, d, d, None, None, 0, 4, 1.0, 0, None)
    assert st == -1 and b"null" in lib.sfa_last_error()


def test_ops_refuse_cpu_tensors_loudly():
    from sink_attention import sink_decode_attention, sink_flash_attention
    q = torch.randn(1, 2, 8, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sink_flash_attention(q, q, q)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sink_decode_attention(q[:, :, :1], q, q)


def test_public_names_match_reference_hot_path():
    import sink_attention
    for name in ("sink_flash_attention", "sink_decode_attention", "patch_verl_with_sink_attention", "unpatch_verl"):
        assert callable(getattr(sink_attention, name))
    import inspect
    sig = inspect.signature(sink_attention.sink_flash_attention)
    assert list(sig.parameters) == ["q", "k", "v", "num_sink", "window_size", "s_aux"]
    assert sig.parameters["num_sink"].default == 4 and sig.parameters["window_size"].default == 512
    assert list(inspect.signature(sink_attention.sink_decode_attention).parameters) == ["q", "k", "v", "s_aux"]


def test_is_packed_truth_table_matches_reference():
    from sink_attention.verl_patch import _is_packed
    truth = G.load("f7_boundary")["packed_truth"].tolist()
    pid_plain = torch.arange(10).view(1, 10)
    pid_packed = torch.tensor([[0, 1, 2, 3, 0, 1, 2, 0, 1, 2]])
    got = [_is_packed(pid_plain), _is_packed(pid_packed), _is_packed(torch.arange(10)),
           _is_packed(torch.zeros(2, 1, dtype=torch.long)), _is_packed(None)]
    assert [int(x) for x in got] == truth


def test_local_s_aux_slicing():
    from sink_attention.verl_patch import _local_s_aux
    s = torch.arange(8.0)
    assert _local_s_aux(None, 4) is None
    assert torch.equal(_local_s_aux(s, 8), s)
    assert torch.equal(_local_s_aux(s, 4), s[:4])      # rank 0 when torch.distributed is not initialised
    assert _local_s_aux(s, 3) is None and _local_s_aux(s[:2], 4) is None


def test_patch_unpatch_and_fallback_routing():
    import transformers.modeling_flash_attention_utils as fa_utils
    from transformers.integrations import flash_attention as fa_int
    import sink_attention.verl_patch as vp
    orig = fa_utils._flash_attention_forward
    calls = []

    def fake_original(*a, **kw):
        calls.append(kw)
        return "fallback"

    fa_utils._flash_attention_forward = fake_original
    try:
        vp.patch_verl_with_sink_attention()
        vp.patch_verl_with_sink_attention()          # idempotent
        assert fa_utils._flash_attention_forward is vp._sink_flash_attention_forward
        assert fa_int._flash_attention_forward is vp._sink_flash_attention_forward
        q = torch.zeros(1, 6, 2, 16)
        sa = torch.zeros(2)
        # every unsupported case goes to the saved original with s_aux put back in kwargs
        for kw in (dict(is_causal=False), dict(attention_mask=torch.ones(1, 6)), dict(softcap=30.0),
                   dict(position_ids=torch.tensor([[0, 1, 2, 0, 1, 2]])),
                   dict(cu_seq_lens_q=torch.tensor([0, 6]), cu_seq_lens_k=torch.tensor([0, 6]), max_length_q=6,
                        max_length_k=6)):
            args = dict(attention_mask=None, is_causal=True)
            args.update(kw)
            mask = args.pop("attention_mask")
            r = fa_utils._flash_attention_forward(q, q, q, mask, 6, s_aux=sa, **args)
            assert r == "fallback" and calls[-1]["s_aux"] is sa
        # supported case reaches the HIP op (which refuses CPU tensors -> proves the routing)
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            fa_utils._flash_attention_forward(q, q, q, None, 6, is_causal=True, sliding_window=4, s_aux=sa)
        with pytest.raises(RuntimeError, match="no CPU fallback"):   # decode route (N_q != N_kv)
            fa_utils._flash_attention_forward(q[:, :1], q, q, None, 1, is_causal=True, s_aux=sa)
        vp.unpatch_verl()
        assert fa_utils._flash_attention_forward is fake_original
        vp.patch_verl_with_sink_attention()          # re-patching works after unpatch
        assert fa_utils._flash_attention_forward is vp._sink_flash_attention_forward
        vp.unpatch_verl()
    finally:
        fa_utils._flash_attention_forward = orig
        fa_int._flash_attention_forward = orig
        vp._original_flash_attention_forward = None


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from sink_attention import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "libsfa.so"))
    with pytest.raises(RuntimeError, match="no Python/CPU fallback"):
        _native.lib()


def test_varlen_bounds_and_default_fallback():
    from sink_attention.varlen import seq_bounds_from_position_ids
    import sink_attention.verl_patch as vp
    pid = torch.tensor([0, 1, 2, 0, 1, 0, 1, 2, 3])
    assert seq_bounds_from_position_ids(pid) == [0, 3, 5, 9]
    assert seq_bounds_from_position_ids(torch.arange(4)) == [0, 4]
    assert vp.ENABLE_VARLEN is False          # default = the reference's fallback behaviour for packed batches


def test_workspace_query_ignores_retired_flag_and_varlen_support_needs_no_gpu():
    from sink_attention import _native
    lib = _native.lib()
    args = (4, 32, 8, 8192, 128, 2, 4, 4096)
    base = lib.sfa_bwd_workspace_bytes(*args, 0)
    # Delta + ds_aux partials + the row constants [B, Hq, 2, N] f32 of the dK/dV kernels; flag bit 0x2 (the retired dS
    # spill) changes nothing
    assert base >= 3 * 4 * 32 * 8192 * 4 and lib.sfa_bwd_workspace_bytes(*args, 0x2) == base
    assert lib.sfa_varlen_supported(2, 128) == 1 and lib.sfa_varlen_supported(1, 64) == 1
    assert lib.sfa_varlen_supported(0, 128) == 0 and lib.sfa_varlen_supported(2, 256) == 0


def test_header_is_plain_c():
    """include/sfa.h must compile as C99 (it is the FFI surface: no C++ / HIP / torch types)."""
    import subprocess
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.NamedTemporaryFile("w", suffix=".c", delete=False) as f:
        f.write('#include "sfa.h"\nint main(void) { sfa_tensor t; t.dtype = SFA_DTYPE_BF16; return t.dtype == SFA_OK; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                        "-fsyntax-only", f.name], capture_output=True, text=True)
    os.unlink(f.name)
    assert r.returncode == 0, r.stderr


def test_hf_option_binding():
    from sink_attention import _hf_args as H
    kw = H.bind((False, 0.1), {"sliding_window": 7, "s_aux": "x"})
    assert kw == {"is_causal": False, "dropout": 0.1, "sliding_window": 7, "s_aux": "x"}
    with pytest.raises(TypeError):
        H.bind((True,), {"is_causal": False})
    with pytest.raises(TypeError):
        H.bind(tuple(range(len(H.OPTIONAL_ORDER) + 1)), {})
    assert H.wants_varlen({"cu_seq_lens_q": 1, "cu_seq_lens_k": 1, "max_length_q": 1, "max_length_k": 1})
    assert not H.wants_varlen({"cu_seq_lens_q": 1})
    # the order is the one of the installed transformers
    import inspect
    import transformers.modeling_flash_attention_utils as fa
    params = list(inspect.signature(fa._flash_attention_forward).parameters)
    tail = [p for p in params[5:] if p in H.OPTIONAL_ORDER]
    assert tail == [p for p in H.OPTIONAL_ORDER if p in tail]


def test_backward_options_map_to_flags():
    """sink_attention.set_backward_options: the overlap flag is the ops' default (the C entry points create nothing unless
    asked), the dK/dV dispatch overrides are exclusive, and the previous setting comes back for restoring"""
    from sink_attention import _native, set_backward_options
    assert _native.bwd_flags() == _native.FLAG_BWD_OVERLAP
    prev = set_backward_options(dkdv="asm")
    assert prev == (True, None) and _native.bwd_flags() == _native.FLAG_BWD_OVERLAP | _native.FLAG_BWD_DKDV_ASM
    assert set_backward_options(overlap=False, dkdv="ws") == (True, "asm")
    assert _native.bwd_flags() == _native.FLAG_BWD_DKDV_WS and _native.bwd_flags(1) == _native.FLAG_BWD_DKDV_WS | 1
    set_backward_options(overlap=True, dkdv="rule")
    assert _native.bwd_flags() == _native.FLAG_BWD_OVERLAP
    # flag values of include/sfa.h
    hdr = open(os.path.join(ROOT, "include", "sfa.h")).read()
    for name, val in (("SFA_FLAG_BWD_OVERLAP", _native.FLAG_BWD_OVERLAP), ("SFA_FLAG_BWD_DKDV_ASM", _native.FLAG_BWD_DKDV_ASM),
                      ("SFA_FLAG_BWD_DKDV_WS", _native.FLAG_BWD_DKDV_WS)):
        assert re.search(r"#define %s 0x%xu" % (name, val), hdr), name
    # the workspace query takes the override bits (the partial-sum area follows the kernel choice)
    lib = _native.lib()
    for f in (0, _native.FLAG_BWD_DKDV_ASM, _native.FLAG_BWD_DKDV_WS, _native.FLAG_BWD_OVERLAP):
        assert lib.sfa_bwd_workspace_bytes(1, 8, 2, 4096, 128, 2, 4, 1024, f) >= 8 * 4096 * 4
