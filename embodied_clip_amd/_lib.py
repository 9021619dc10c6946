"""ctypes binding of the C-ABI in include/ec_amd.h.

The product path has NO CPU fallback: if the HIP library is missing this
module raises, loudly.  (The CPU oracle under oracle/ is test infrastructure
and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# EC_AMD_LIB: another BUILD of this library (same C-ABI; same-box A/B of two builds, tools/ab.sh) -- still a HIP library, never a fallback
LIB_PATH = os.environ.get("EC_AMD_LIB") or os.path.join(_HERE, "lib", "libec_amd.so")

c_void_p, c_int, c_size_t, c_float = C.c_void_p, C.c_int, C.c_size_t, C.c_float

# name -> (restype, argtypes); must list every symbol include/ec_amd.h declares
# (tests/test_cabi_symbols.py parses the header and checks both directions).
SIGNATURES = {
    "ec_version": (c_int, []),
    "ec_strerror": (C.c_char_p, [c_int]),
    "ec_bind_streams": (c_int, [c_void_p, c_int, c_int]),
    "ec_stream_pair_overlap": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "ec_conv_bf16": (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    "ec_conv_bf16_ld": (c_int, [c_void_p] * 5 + [c_int] * 9 + [c_void_p]),
    "ec_conv3x3_img_pack": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "ec_conv3x3_img_bf16": (c_int, [c_void_p] * 4 + [c_int] * 5 + [c_void_p]),
    "ec_conv3x3_img_bf16_ld": (c_int, [c_void_p] * 4 + [c_int] * 6 + [c_void_p]),
    "ec_bneck3_packed_elems": (c_size_t, [c_int]),
    "ec_bneck3_pack_weights": (c_int, [c_void_p] * 4 + [c_int, c_void_p]),
    "ec_bneck_conv123_bf16": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_void_p]),
    "ec_bneck_packed_elems": (c_size_t, [c_int]),
    "ec_bneck_pack_weights": (c_int, [c_void_p] * 3 + [c_int, c_void_p]),
    "ec_bneck_conv23_bf16": (c_int, [c_void_p] * 6 + [c_int] * 4 + [c_void_p]),
    "ec_clip_resize_table_ints": (c_size_t, [c_int, c_int, c_int]),
    "ec_clip_resize_table": (c_int, [c_int, c_int, c_int, c_void_p, c_size_t]),
    "ec_clip_resize_crop_u8": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ec_gemm_bf16": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "ec_split3_bf16": (c_int, [c_void_p] * 2 + [C.c_long, c_int, c_void_p]),
    "ec_gemm_bf16a_x3": (c_int, [c_void_p] * 4 + [C.c_long, c_int, c_int, c_int, c_void_p]),
    "ec_dw_tn_x3_splits": (c_int, [C.c_long, c_int]),
    "ec_dw_tn_x3": (c_int, [c_void_p] * 4 + [C.c_long, c_int, c_void_p]),
    "ec_stem_conv1": (c_int, [c_void_p] * 4 + [c_int] * 4 + [c_void_p]),
    "ec_avgpool2_bf16": (c_int, [c_void_p] * 2 + [c_int] * 4 + [c_void_p]),
    "ec_avgpool2_bf16_ld": (c_int, [c_void_p] * 2 + [c_int] * 5 + [c_void_p]),
    "ec_nhwc_bf16_to_nchw_f32": (c_int, [c_void_p] * 2 + [c_int] * 3 + [c_void_p]),
    "ec_nchw_f32_to_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ec_spatial_mean_bf16": (c_int, [c_void_p] * 2 + [c_int] * 3 + [c_void_p]),
    "ec_rn50_create": (c_int, [C.POINTER(c_void_p), c_int, C.POINTER(c_int), c_int, c_void_p, c_void_p, c_size_t,
                               c_void_p, c_size_t]),
    "ec_rn50tv_create": (c_int, [C.POINTER(c_void_p), C.POINTER(c_int), c_int, c_void_p, c_void_p, c_size_t, c_void_p,
                                 c_size_t]),
    "ec_conv_bf16_s2": (c_int, [c_void_p] * 5 + [c_int] * 7 + [c_void_p]),
    "ec_stem7_pool": (c_int, [c_void_p, c_int, C.POINTER(c_float), C.POINTER(c_float), c_void_p, c_void_p, c_void_p,
                              c_int, c_int, c_int, c_void_p]),
    "ec_rn50_destroy": (None, [c_void_p]),
    "ec_rn50_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "ec_rn50_out_channels": (c_int, [c_void_p]),
    "ec_rn50_out_spatial": (c_int, [c_void_p]),
    "ec_rn50_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_int, c_void_p]),
    "ec_rn50_num_ops": (c_int, [c_void_p]),
    "ec_rn50_plan_hash": (C.c_uint64, [c_void_p]),
    "ec_rn50_set_conv8_min_tiles": (c_int, [c_void_p, c_int]),
    "ec_text_create": (c_int, [C.POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p,
                               c_size_t]),
    "ec_text_destroy": (None, [c_void_p]),
    "ec_text_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "ec_text_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "ec_probe_head": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p]),
    "ec_probe_pool3": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ec_conv1x1_pair_pool_bf16": (c_int, [c_void_p] * 9 + [c_int] * 6 + [c_void_p]),
    "ec_conv1x1_pair_bf16": (c_int, [c_void_p] * 11 + [C.c_long, c_int, c_int, c_int, c_void_p]),
    "ec_stem_conv1_u8": (c_int, [c_void_p, C.POINTER(c_float), C.POINTER(c_float), c_void_p, c_void_p, c_void_p]
                         + [c_int] * 4 + [c_void_p]),
    "ec_rn50_forward_u8": (c_int, [c_void_p, c_void_p, C.POINTER(c_float), C.POINTER(c_float), c_int, c_void_p,
                                   c_size_t, c_void_p, c_int, c_void_p]),
    "ec_gemm_f32": (c_int, [c_void_p] * 3 + [c_int] * 3 + [C.c_long] * 4 + [c_int, c_int] + [c_void_p] * 3 + [c_int]
                    + [c_void_p] * 2 + [c_int, c_void_p]),
    "ec_policy_create": (c_int, [C.POINTER(c_void_p), c_void_p]),
    "ec_policy_destroy": (None, [c_void_p]),
    "ec_policy_num_param_tensors": (c_int, [c_void_p]),
    "ec_policy_flat_size": (c_size_t, [c_void_p]),
    "ec_policy_param_offset": (c_int, [c_void_p, c_int, C.POINTER(c_size_t), C.POINTER(c_size_t)]),
    "ec_policy_workspace_bytes": (c_size_t, [c_void_p, c_int, c_int, c_int]),
    "ec_policy_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                  c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_void_p]),
    "ec_policy_forward2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                   c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_void_p]),
    "ec_policy_act": (c_int, [c_void_p] * 4 + [c_int] + [c_void_p] * 3 + [c_int, c_void_p, c_size_t, c_int] + [c_void_p] * 5
                      + [C.c_uint64, C.c_uint64, c_int, c_void_p]),
    "ec_policy_backward2": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_size_t,
                                    c_void_p, c_void_p, c_void_p, c_void_p]),
    "ec_policy_backward3": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_size_t,
                                    c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ec_policy_set_goal_table": (c_int, [c_void_p, c_void_p]),
    "ec_policy_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_size_t,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    "ec_gae": (c_int, [c_void_p] * 7 + [c_int, c_int, c_float, c_float, c_float, c_void_p]),
    "ec_ppo_loss": (c_int, [c_void_p] * 8 + [C.c_long, c_int, c_float, c_float, c_float, c_float, c_void_p]),
    "ec_ppo_loss_ex": (c_int, [c_void_p] * 8 + [C.c_long, c_int, c_float, c_float, c_float, c_float, c_float, c_void_p]),
    "ec_sample_actions": (c_int, [c_void_p] * 4 + [c_int, c_int, C.c_uint64, C.c_uint64, c_int, c_void_p]),
    "ec_vit_create": (c_int, [C.POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t, c_void_p,
                              c_size_t]),
    "ec_vit_destroy": (None, [c_void_p]),
    "ec_vit_tokens": (c_int, [c_void_p]),
    "ec_vit_set_conv8_min_tiles": (c_int, [c_void_p, c_int]),
    "ec_vit_plan_hash": (C.c_uint64, [c_void_p]),
    "ec_vit_workspace_bytes": (c_size_t, [c_void_p, c_int]),
    "ec_vit_forward": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "ec_bf16_to_f32": (c_int, [c_void_p, c_void_p, C.c_long, C.c_long, C.c_long, c_void_p]),
    "ec_attnpool_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "ec_attnpool_forward": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int] + [c_void_p] * 8 + [c_size_t, c_void_p,
                                    c_void_p]),
    "ec_clip_adam_scratch_doubles": (c_int, []),
    "ec_clip_adam_step": (c_int, [c_void_p] * 5 + [C.c_long, c_float, c_float, c_float, c_float, c_float, c_int,
                                  c_void_p]),
}


class PolicyCfg(C.Structure):
    _fields_ = [(n, c_int) for n in ("in_channels", "spatial", "hidden", "goal_dims", "num_goals", "num_actions",
                                     "compress_hid", "compress_out", "comb_hid", "comb_out", "fusion", "dual")]

_lib = None


class EcError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libec_amd.so (built by ``__graft_entry__.build()`` / ``make``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"embodied_clip_amd: HIP library not found at {LIB_PATH}. Build it with `make` or "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, gfx950). "
            "There is deliberately no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError here == header/library drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ec_strerror(rc).decode()
        raise EcError(f"{what or 'ec_amd call'} failed: {msg} (code {rc})")


def ptr(t) -> int:
    """Device pointer of a torch tensor (or None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def stream_ptr(device=None) -> int:
    """Raw ``hipStream_t`` of torch's current stream on ``device`` (default: the current device)."""
    import torch
    return torch.cuda.current_stream(device).cuda_stream


def concurrent_streams(n: int, device, spin_us: int = 200, max_candidates: int = 24):
    """``n`` new torch streams on ``device`` that really run concurrently.  The HIP runtime binds a stream to one of its (four)
    hardware queues at the stream's FIRST submission, and two streams that land on one queue serialise (measured: every second
    two-slice worker of a process ran its two encoder launches one after the other, 48 k instead of 63 k env-frames/s).  The set
    is built greedily: a candidate stream gets its first work while the accepted ones are busy (``ec_bind_streams``), is then
    measured against every accepted stream (``ec_stream_pair_overlap``: both busy / one alone ~ 1 when concurrent, ~ 2 when
    serialised) and joins the set only if it overlaps with all of them; rejected candidates stay alive until the set is complete,
    so the runtime cannot hand their queue out again.  This finds ``n`` distinct queues whatever other live streams of the process
    already sit on them.  Raises if ``max_candidates`` do not yield the set (``n`` above the number of hardware queues): two
    launches in flight is what the engine's numbers rest on, and a silent fall-back to serial streams would misreport them."""
    import torch
    lib = load()
    dev = torch.device(device)
    accepted, rejected, worst = [], [], 0.0
    with torch.cuda.device(dev):
        for _ in range(max_candidates):
            if len(accepted) == n:
                break
            cand = torch.cuda.Stream(device=dev)
            group = accepted + [cand]
            arr = (C.c_void_p * len(group))(*[s.cuda_stream for s in group])
            check(lib.ec_bind_streams(arr, len(group), spin_us), "ec_bind_streams")
            ok = True
            for s in accepted:
                r = C.c_float()
                check(lib.ec_stream_pair_overlap(s.cuda_stream, cand.cuda_stream, spin_us, C.byref(r)), "ec_stream_pair_overlap")
                if r.value >= 1.5:
                    ok, worst = False, max(worst, r.value)
                    break
            (accepted if ok else rejected).append(cand)
    if len(accepted) < n and os.environ.get("EC_STREAMS_UNVERIFIED") == "1":
        # counter-collecting profiler runs only (tools/pmc_collect.sh: rocprofv3 --pmc serialises every dispatch, so no pair of streams can pass the
        # check there, and per-kernel counters do not depend on it); bench.py refuses to run with this set
        import sys
        print(f"embodied_clip_amd: EC_STREAMS_UNVERIFIED=1 -- using {n - len(accepted)} stream(s) that FAILED the concurrency check "
              f"(both-busy / alone = {worst:.2f}); timings of this process say nothing about the two-stream engine", file=sys.stderr)
        return accepted + rejected[:n - len(accepted)]
    if len(accepted) < n:
        raise RuntimeError(f"could not obtain {n} concurrent HIP streams on {dev}: {len(accepted)} found among {max_candidates} candidates "
                           f"(both-busy / alone of a rejected pair = {worst:.2f})")
    return accepted


def on_device(fn):
    """Method decorator: run ``fn`` with ``self.device`` as the current HIP device, so that the kernels behind the
    C-ABI launch on the GPU that owns the handle's tensors (AllenAct places preprocessors on devices other than the
    process default) and ``stream_ptr()`` resolves to that GPU's current stream."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        import torch
        dev = getattr(self, "device", None)
        if dev is None or dev.type != "cuda" or dev.index is None or dev.index == torch.cuda.current_device():
            return fn(self, *a, **kw)
        with torch.cuda.device(dev):
            return fn(self, *a, **kw)
    return wrapped


def tensor_guard(t):
    """Context manager making ``t``'s GPU the current device (no-op on the current one)."""
    import contextlib
    import torch
    if t is None or not t.is_cuda or t.device.index == torch.cuda.current_device():
        return contextlib.nullcontext()
    return torch.cuda.device(t.device)
