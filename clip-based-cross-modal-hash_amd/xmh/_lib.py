"""ctypes binding of libxmh.so (the C ABI declared in include/xmh.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C
clip-based-cross-modal-hash_amd`` with ``hipcc --offload-arch=gfx950``.  If it
is missing this module raises at import: the product path never falls back to
PyTorch/CPU arithmetic.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XMH_LIB", os.path.join(_HERE, "libxmh.so"))

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "libxmh.so not found at %s -- build it first (python -c 'import __graft_entry__ as g; g.build()' "
        "or make -C clip-based-cross-modal-hash_amd). There is no CPU fallback." % LIB_PATH)

# PyTorch first: it ships its own HIP runtime, and the process must end up with ONE.  Loaded after torch, libxmh.so binds to the runtime
# torch already brought in (same soname); loaded before it, the library initialises /opt/rocm's copy, torch then initialises its own, and
# the first caller of the second one finds "no ROCm-capable device" (seen with build() and smoke() in one process).  Every caller of this
# module hands it torch tensors anyway.
import torch  # noqa: E402,F401

lib = C.CDLL(LIB_PATH)

vp, i64, i32, sz = C.c_void_p, C.c_int64, C.c_int, C.c_size_t


class ScanPlan(C.Structure):
    _fields_ = [("chunk", i64), ("nchunk", i64), ("nqtile", i64), ("qpad", i64), ("nbuckets", i64), ("ws_bytes", sz)]


class Linear(C.Structure):                     # xmh_linear
    _fields_ = [("w_f32", vp), ("w_hi", vp), ("w_lo", vp), ("bias", vp), ("n", i64), ("k", i64)]


class ClipBlock(C.Structure):                  # xmh_clip_block
    _fields_ = [("ln1_w", vp), ("ln1_b", vp), ("ln2_w", vp), ("ln2_b", vp),
                ("qkv", Linear), ("out", Linear), ("fc", Linear), ("proj", Linear)]


class VitWeights(C.Structure):                 # xmh_vit_weights
    _fields_ = [("resolution", i32), ("patch", i32), ("width", i32), ("heads", i32), ("layers", i32), ("out_dim", i32),
                ("conv1", Linear), ("cls", vp), ("pos", vp), ("ln_pre_w", vp), ("ln_pre_b", vp), ("ln_post_w", vp),
                ("ln_post_b", vp), ("proj", Linear), ("blocks", C.POINTER(ClipBlock))]


class TextWeights(C.Structure):                # xmh_text_weights
    _fields_ = [("vocab", i32), ("context", i32), ("width", i32), ("heads", i32), ("layers", i32), ("out_dim", i32),
                ("tok_emb", vp), ("pos", vp), ("ln_final_w", vp), ("ln_final_b", vp), ("proj", Linear),
                ("blocks", C.POINTER(ClipBlock))]


class DcmhtHead(C.Structure):                  # xmh_dcmht_head
    _fields_ = [("v_proj", Linear), ("out_proj", Linear), ("norm_is_batchnorm", i32), ("norm_eps", C.c_float),
                ("norm_w", vp), ("norm_b", vp), ("bn_mean", vp), ("bn_var", vp), ("fc2", Linear)]


class MithMlp(C.Structure):                    # xmh_mith_mlp
    _fields_ = [("ln_w", vp), ("ln_b", vp), ("ln_eps", C.c_float), ("fc1", Linear), ("fc2", Linear)]


class MithHead(C.Structure):                   # xmh_mith_head
    _fields_ = [("width", i32), ("k_bits", i32), ("top_k", i32), ("res_layers", i32), ("layers", i32), ("heads", i32),
                ("mlps", C.POINTER(MithMlp)), ("concept", Linear), ("pos_enc", vp), ("blocks", C.POINTER(ClipBlock)),
                ("hash_w", vp), ("hash_b", vp)]


# name -> (restype, argtypes); mirrors include/xmh.h one to one
PROTOTYPES = {
    "xmh_version": (i32, []),
    "xmh_build_id": (C.c_char_p, []),
    "xmh_last_error": (C.c_char_p, []),
    "xmh_prof_enable": (i32, [i32]),
    "xmh_range_push": (i32, [C.c_char_p]),
    "xmh_range_pop": (i32, []),
    "xmh_scan_verify": (i32, [i32]),
    "xmh_prof_read": (i32, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(i64)]),
    "xmh_pack_sign": (i32, [vp, i64, i32, vp, vp, vp, vp, vp]),
    "xmh_pack_pair_argmax": (i32, [vp, i64, i32, vp, vp, vp]),
    "xmh_unpack_pm1": (i32, [vp, vp, i64, i32, vp, vp]),
    "xmh_pack_labels": (i32, [vp, i32, i64, i32, vp, vp]),
    "xmh_hamming_dist": (i32, [vp, vp, vp, vp, i64, i64, i32, vp, vp, vp]),
    "xmh_label_sim": (i32, [vp, vp, i64, i64, i32, vp, vp]),
    "xmh_scan_plan_make": (i32, [i64, i64, i32, i32, C.POINTER(ScanPlan)]),
    "xmh_scan_pair_cache_bytes": (sz, [i64, i64, i32, i32]),
    "xmh_scan_pair_cache_offset": (sz, [i64, i64, i32, i32]),
    "xmh_scan_ws_bytes_nocache": (sz, [i64, i64, i32, i32]),
    "xmh_scan_describe": (i32, [i64, i64, i32, i32, i32, C.c_char_p, sz]),
    "xmh_hamming_hist": (i32, [vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, vp, sz, vp, vp, vp]),
    "xmh_hamming_ap": (i32, [vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, vp, sz, vp, vp, vp, i64, vp, vp, vp]),
    "xmh_hamming_map": (i32, [vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, vp, sz, i64, vp, vp, vp, vp]),
    "xmh_scan_totals_offset": (sz, [i64, i64, i32, i32, C.POINTER(C.c_size_t)]),
    "xmh_hamming_map_sharded": (i32, [vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, vp, sz, vp, i32, i32, i64, vp, vp, vp, vp]),
    "xmh_shard_slice_offsets": (i32, [vp, i32, i32, i32, vp, vp]),
    "xmh_hamming_map_sharded_offsets": (i32, [vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, vp, sz, vp, i32, i64, vp, vp, vp, vp]),
    "xmh_calc_map_k_ws_bytes": (sz, [i64, i64, i32, i32]),
    "xmh_calc_map_k": (i32, [vp, vp, vp, vp, i64, i64, i32, i32, i64, vp, sz, C.POINTER(C.c_double), C.POINTER(i32), vp]),
    "xmh_map_finalize": (i32, [vp, vp, i64, vp, vp]),
    "xmh_shard_offsets": (i32, [vp, i32, i32, i64, i32, vp, vp, vp, vp]),
    "xmh_gemm_nt_f32": (i32, [vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, i64, i64, i32, i32, vp]),
    "xmh_gemm_nt_h16": (i32, [vp, i64, vp, i64, vp, vp, i64, vp, i64, i64, i64, i64, i32, vp]),
    "xmh_gemm_nt_split16": (i32, [vp, i64, vp, vp, i64, vp, vp, i64, vp, i64, i64, i64, i64, i32, vp]),
    "xmh_cast_f32_to_f16": (i32, [vp, vp, i64, vp]),
    "xmh_layernorm_f32": (i32, [vp, i64, vp, vp, C.c_float, vp, i64, i64, i32, vp]),
    "xmh_attention_f32": (i32, [vp, i64, i32, i32, i32, i32, vp, vp, vp]),
    "xmh_attention_split16": (i32, [vp, i64, i32, i32, i32, i32, vp, vp, vp]),
    "xmh_image_preprocess_u8": (i32, [vp, i64, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
    "xmh_im2col_patch": (i32, [vp, i64, i32, i32, i32, vp, vp]),
    "xmh_vit_assemble": (i32, [vp, vp, vp, vp, vp, C.c_float, vp, i64, i32, i32, vp]),
    "xmh_text_embed": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]),
    "xmh_gather_rows": (i32, [vp, i64, vp, i32, i32, vp, i64, i32, vp]),
    "xmh_affine_cols": (i32, [vp, vp, vp, vp, vp, C.c_float, vp, i64, i32, vp]),
    "xmh_pair_softmax": (i32, [vp, vp, i64, i32, vp]),
    "xmh_lta_aggregate": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]),
    "xmh_bitwise_hash": (i32, [vp, vp, vp, vp, vp, i64, i32, i32, vp]),
    "xmh_clip_workspace_bytes": (sz, [i64, i32, i32, i32, i32, i32]),
    "xmh_clip_blocks_forward": (i32, [C.POINTER(ClipBlock), i32, i32, i32, vp, i64, i32, i32, vp, i32, vp, sz, vp]),
    "xmh_clip_saved_bytes": (sz, [i64, i32, i32, i32]),
    "xmh_clip_blocks_forward_saved": (i32, [C.POINTER(ClipBlock), i32, i32, i32, vp, i64, i32, i32, vp, i32, vp, sz, vp, sz, vp]),
    "xmh_vit_b32_forward": (i32, [C.POINTER(VitWeights), vp, i64, i32, vp, vp, vp, sz, vp]),
    "xmh_text_forward": (i32, [C.POINTER(TextWeights), vp, vp, i64, i32, i32, vp, vp, vp, vp, sz, vp]),
    "xmh_text_forward_packed": (i32, [C.POINTER(TextWeights), vp, vp, i64, i64, i32, i32, vp, vp, sz, vp]),
    "xmh_text_forward_packed_dev": (i32, [C.POINTER(TextWeights), vp, vp, i64, i32, i32, vp, vp, vp, sz, vp]),
    "xmh_head_workspace_bytes": (sz, [i64, i32, i32]),
    "xmh_head_dcmht": (i32, [C.POINTER(DcmhtHead), vp, i64, i32, vp, vp, vp, vp, sz, vp]),
    "xmh_head_dsph": (i32, [C.POINTER(Linear), vp, i64, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
    "xmh_head_mith_workspace_bytes": (sz, [i64, i32, i32, i32, i32]),
    "xmh_head_mith": (i32, [C.POINTER(MithHead), vp, vp, vp, i64, i32, i32, vp, vp, vp, sz, vp]),
    "xmh_row_l2normalize": (i32, [vp, i64, i32, vp, vp, vp]),
    "xmh_pairwise_l2_from_gram": (i32, [vp, vp, vp, i64, i64, vp]),
    "xmh_affine_inplace": (i32, [vp, i64, C.c_float, C.c_float, vp]),
    "xmh_gemm_f32_sort_ws_bytes": (sz, [i64, i64]),
    "xmh_gemm_f32_sort_map": (i32, [vp, vp, vp, vp, i64, i64, i32, i32, i64, vp, sz, vp, vp, vp, vp]),
    "xmh_float_sort_ap": (i32, [vp, vp, vp, i64, i64, i32, i64, vp, sz, vp, vp, vp]),
    "xmh_pair_similarity_loss": (i32, [vp, vp, i64, i32, vp, i32, i32, C.c_float, C.c_float, vp, vp]),
    "xmh_quant_loss": (i32, [vp, i64, vp, vp]),
    "xmh_pair_similarity_loss_grad": (i32, [vp, vp, i64, i32, vp, i32, i32, C.c_float, C.c_float, C.c_float, vp, vp, i32, vp]),
    "xmh_quant_loss_grad": (i32, [vp, i64, C.c_float, vp, vp, i32, vp]),
    "xmh_topk_ws_bytes": (sz, [i64, i64, i32, i32]),
    "xmh_hamming_topk": (i32, [vp, vp, i64, i64, i32, i32, i64, vp, sz, vp, vp, vp]),
    "xmh_topk_ws_init": (i32, [i64, i64, i32, i32, vp, sz, vp]),
    "xmh_hamming_topk_prepared": (i32, [vp, vp, i64, i64, i32, i32, i64, vp, sz, vp, vp, vp]),
    "xmh_topk_ternary_ws_bytes": (sz, [i64, i64, i32, i32]),
    "xmh_topk_ternary_ws_init": (i32, [i64, i64, i32, i32, vp, sz, vp]),
    "xmh_hamming_topk_ternary": (i32, [vp, vp, vp, vp, i64, i64, i32, i32, i64, vp, sz, i32, vp, vp, vp]),
    "xmh_topk_describe": (i32, [i64, i64, i32, i32, C.c_char_p, sz]),
    "xmh_topk_record_bytes": (sz, [i64, i32]),
    "xmh_topk_merge_host": (i32, [vp, i32, i64, i32, vp, vp]),
}

for _name, (_res, _args) in PROTOTYPES.items():
    _fn = getattr(lib, _name)          # AttributeError here == header and library out of sync
    _fn.restype = _res
    _fn.argtypes = _args


class XmhError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    """Map a non-zero C status to RuntimeError(xmh_last_error()) (SURVEY 8b 'Errors')."""
    if rc != 0:
        msg = lib.xmh_last_error().decode("utf-8", "replace")
        raise XmhError("%s failed (%d): %s" % (what or "libxmh call", rc, msg))


def ptr(t):
    """Device (or host) pointer of a torch tensor / None -> c_void_p."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def current_stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def prof_enable(on=True) -> None:
    """True / 1: HIP events around the dominant kernels (prof_read); 2: roctx ranges around every phase of the path (a
    `rocprofv3 --marker-trace` timeline gets phase markers); 3: both; False / 0: off."""
    check(lib.xmh_prof_enable(int(on)), "xmh_prof_enable")


def scan_verify(on: bool = True) -> None:
    """debug mode: every unsharded ranking call is re-derived with the masked VALU kernels and compared (xmh_scan_verify); a
    disagreement raises RuntimeError from the call.  About 3x the cost of an evaluation plus a stream synchronisation."""
    check(lib.xmh_scan_verify(int(bool(on))), "xmh_scan_verify")


class prof_range:
    """roctx range from the host layer (collectives of the sharded evaluation, the encode loop): ``with prof_range("name"):``.
    A no-op -- two foreign calls that return at once -- while ranges are off (prof_enable(2))."""

    def __init__(self, name: str):
        self.name = name.encode()

    def __enter__(self):
        lib.xmh_range_push(self.name)
        return self

    def __exit__(self, *exc):
        lib.xmh_range_pop()
        return False


def prof_read(name: str):
    """-> (mean launch ms, launches) of the named kernel since prof_enable(True)."""
    ms, n = C.c_double(0.0), i64(0)
    check(lib.xmh_prof_read(name.encode(), C.byref(ms), C.byref(n)), "xmh_prof_read")
    return ms.value, n.value
