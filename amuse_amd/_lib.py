"""ctypes binding of libamuse_hip.so (include/amuse_hip.h).

There is NO CPU fallback: if the HIP library is missing or fails to load, importing a symbol from
here raises.  (The oracle under oracle/ is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
# AMUSE_HIP_LIB: load another build of the same C ABI (kernel A/B measurements, tools/build_variant.sh)
LIB_PATH = Path(os.environ.get("AMUSE_HIP_LIB") or _HERE / "libamuse_hip.so")

PREC_F32, PREC_BF16, PREC_F32X, PREC_F16 = 0, 1, 2, 3
UPD_F32, UPD_BF16, UPD_ENCODER, UPD_F32X, UPD_F16, UPD_ALL = 1, 2, 4, 8, 16, 31
QUAT_P3D, QUAT_LEGACY = 0, 1
ABI_VERSION = 5
ARCH_ENC, ARCH_DEC, ARCH_ENC_POSE, ARCH_DEC_POSE = 0, 1, 2, 3   # include/amuse_hip.h AMUSE_ARCH_*

EXPORTS = [
    "amuse_abi_version", "amuse_last_error", "amuse_create", "amuse_update_weights", "amuse_update_weights_device", "amuse_destroy", "amuse_set_schedule",
    "amuse_sample", "amuse_denoise_step", "amuse_diffusion_forward", "amuse_vae_decode", "amuse_vae_encode", "amuse_smplx_to_feats", "amuse_diffusion_backward",
    "amuse_counter_normal", "amuse_set_clips_per_group", "amuse_set_decode_path", "amuse_plan", "amuse_debug_last_plan", "amuse_profile_sample",
    "amuse_audio_create", "amuse_audio_destroy", "amuse_audio_fbank", "amuse_audio_encode", "amuse_audio_features",
    "amuse_debug_gemm",
    "amuse_debug_tile", "amuse_debug_f16_split", "amuse_debug_set_decode_tap",
    "amuse_create_arch", "amuse_denoiser_param_count", "amuse_arch", "amuse_state_dim", "amuse_denoise_step_pose", "amuse_feats_to_smplx",
    "amuse_debug_set_ablation",
    "amuse_train_ws_floats", "amuse_train_set_lane", "amuse_train_ln_fwd", "amuse_train_ln_bwd", "amuse_train_bias_gelu_drop_fwd", "amuse_train_bias_gelu_drop_bwd", "amuse_train_colsum",
    "amuse_train_layer_fwd", "amuse_train_layer_bwd", "amuse_train_linear_fwd", "amuse_train_linear_bwd", "amuse_train_adamw", "amuse_train_adamw_dev", "amuse_train_epoch_advance", "amuse_train_epoch_set", "amuse_train_attn_fwd", "amuse_train_attn_bwd",
]


class AmuseHipError(RuntimeError):
    pass


_FP = C.c_void_p


class TrainLayer(C.Structure):
    """include/amuse_hip.h `amuse_train_layer` (device addresses as integers: tensor.data_ptr())."""
    _fields_ = ([("rows", C.c_long), ("B", C.c_int), ("S", C.c_int), ("H", C.c_int), ("ff", C.c_int), ("p", C.c_float), ("p_attn", C.c_float),
                 ("seed", C.c_uint64), ("off", C.c_uint64 * 5)]
                + [(n, _FP) for n in ("Wo bo g1 be1 Wv bv Wc bc g2 be2 W1 b1 W2 b2 g3 be3 x o2 mem x1 zh1 r1 c vk xm zh2 r2 h a out zh3 r3 tmp dout dx do2 dmem "
                                      "dWo dbo dg1 dbe1 dWv dbv dWc dbc dg2 dbe2 dW1 db1 dW2 db2 dg3 dbe3 s128a s128b s512a s512b sdc ws Win bin qkv lse dqkv dWin dbin").split()]
                + [("off_self", C.c_uint64)])


class Schedule(C.Structure):
    _fields_ = [("n_steps", C.c_int), ("timesteps", C.POINTER(C.c_int)), ("coef", C.POINTER(C.c_float)),
                ("freqs", C.POINTER(C.c_float))]


_lib = None


def load() -> C.CDLL:
    """Load the shared library and declare every prototype.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch FIRST: it ships its own HIP runtime (torch/lib/libamdhip64.so).  Loaded behind this library - which links the system's - the process would hold two
    # runtimes, and the second one sees no device ("no ROCm-capable device is detected" from amuse_create in a process that called __graft_entry__.build(), which
    # loads the library, before anything imported torch).  With torch's copy resident the library binds to it.  A caller without torch (a plain ctypes user of
    # include/amuse_hip.h through this loader) has one runtime anyway: the import is attempted, not required ("Linking").
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not LIB_PATH.exists():
        raise AmuseHipError(
            f"{LIB_PATH} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C amuse_amd/csrc`.  amuse_amd has no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    vp, fp, ip = C.c_void_p, C.c_void_p, C.POINTER(C.c_int)  # device pointers travel as void*
    u64 = C.c_uint64
    lib.amuse_abi_version.restype = C.c_int
    lib.amuse_last_error.restype = C.c_char_p
    lib.amuse_create.restype = vp
    lib.amuse_create.argtypes = [C.c_int, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float), C.c_size_t]
    lib.amuse_create_arch.restype = vp
    lib.amuse_create_arch.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float), C.c_size_t]
    lib.amuse_denoiser_param_count.restype = C.c_size_t
    lib.amuse_denoiser_param_count.argtypes = [C.c_int]
    lib.amuse_arch.restype = C.c_int
    lib.amuse_arch.argtypes = [vp]
    lib.amuse_state_dim.restype = C.c_size_t
    lib.amuse_state_dim.argtypes = [vp]
    lib.amuse_denoise_step_pose.restype = C.c_int
    lib.amuse_denoise_step_pose.argtypes = [vp, fp, C.c_int, fp, fp, fp, ip, C.c_int, C.c_int, fp, vp]
    lib.amuse_feats_to_smplx.restype = C.c_int
    lib.amuse_feats_to_smplx.argtypes = [vp, fp, C.c_int, C.c_int, fp, fp, vp]
    lib.amuse_debug_set_ablation.restype = C.c_int
    lib.amuse_debug_set_ablation.argtypes = [vp, C.c_int]
    lib.amuse_update_weights.restype = C.c_int
    lib.amuse_update_weights.argtypes = [vp, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float), C.c_size_t, C.c_int, vp]
    lib.amuse_update_weights_device.restype = C.c_int
    lib.amuse_update_weights_device.argtypes = [vp, vp, vp, C.c_int, vp]
    lib.amuse_destroy.restype = None
    lib.amuse_destroy.argtypes = [vp]
    lib.amuse_set_schedule.argtypes = [vp, C.POINTER(Schedule), vp]
    lib.amuse_sample.argtypes = [vp, fp, fp, fp, C.c_int, C.c_int, u64, u64, fp, fp, fp, fp, vp]
    lib.amuse_denoise_step.argtypes = [vp, fp, C.c_int, fp, fp, fp, C.c_int, C.c_int, fp, fp, vp]
    lib.amuse_diffusion_forward.argtypes = [vp, fp, fp, ip, C.POINTER(C.c_float), C.POINTER(C.c_float), fp, fp, fp, C.c_int, C.c_int, fp, fp, vp]
    lib.amuse_diffusion_forward.restype = C.c_int
    lib.amuse_vae_decode.argtypes = [vp, fp, ip, C.c_int, C.c_int, C.c_int, fp, fp, fp, vp]
    lib.amuse_vae_encode.argtypes = [vp, fp, ip, C.c_int, C.c_int, fp, fp, fp, fp, vp]
    lib.amuse_smplx_to_feats.argtypes = [vp, fp, fp, C.c_int, fp, vp]
    lib.amuse_diffusion_backward.argtypes = [vp, fp, fp, fp, C.c_int, C.c_int, C.c_int, u64, u64, fp, fp, fp, fp, fp, vp]
    lib.amuse_counter_normal.argtypes = [vp, u64, u64, C.c_int, C.c_int, C.c_int, fp, vp]
    lib.amuse_set_clips_per_group.argtypes = [vp, C.c_int]
    lib.amuse_set_decode_path.argtypes = [vp, C.c_int]
    lib.amuse_profile_sample.argtypes = [vp, fp, fp, fp, C.c_int, C.c_int, C.c_int, fp, vp]
    lib.amuse_plan.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, ip, ip, ip, ip]
    lib.amuse_debug_last_plan.argtypes = [vp, ip, ip, ip, ip]
    lib.amuse_plan.restype = lib.amuse_debug_last_plan.restype = C.c_int
    for n in ("amuse_set_schedule", "amuse_sample", "amuse_denoise_step", "amuse_diffusion_forward", "amuse_vae_decode", "amuse_vae_encode", "amuse_smplx_to_feats",
              "amuse_diffusion_backward", "amuse_counter_normal", "amuse_set_clips_per_group", "amuse_set_decode_path",
              "amuse_profile_sample"):
        getattr(lib, n).restype = C.c_int
    lib.amuse_audio_create.restype = vp
    lib.amuse_audio_create.argtypes = [C.c_int, fp, fp, fp, C.c_size_t, fp, fp, C.c_float, C.c_float, C.c_int]
    lib.amuse_audio_destroy.argtypes = [vp]
    lib.amuse_audio_destroy.restype = None
    lib.amuse_audio_fbank.argtypes = [vp, fp, C.c_int, C.c_int, fp, vp]
    lib.amuse_audio_encode.argtypes = [vp, C.c_int, fp, C.c_int, fp, fp, C.c_int, vp]
    lib.amuse_audio_features.argtypes = [vp, fp, C.c_int, C.c_int, fp, fp, fp, vp]
    lib.amuse_debug_gemm.argtypes = [vp, vp, fp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]
    lib.amuse_debug_tile.argtypes = [vp, vp, C.c_int, C.c_int, C.c_int, vp]
    lib.amuse_debug_f16_split.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_uint16), C.POINTER(C.c_uint16)]
    lib.amuse_debug_f16_split.restype = C.c_int
    lib.amuse_debug_set_decode_tap.argtypes = [vp, fp]
    lib.amuse_debug_set_decode_tap.restype = C.c_int
    # training-step glue (csrc/k_train.hip): raw device addresses (tensor.data_ptr()) as void pointers
    u64 = C.c_uint64
    lib.amuse_train_ws_floats.argtypes = []
    lib.amuse_train_ws_floats.restype = C.c_size_t
    lib.amuse_train_set_lane.argtypes = [C.c_int]
    lib.amuse_train_set_lane.restype = C.c_int
    lib.amuse_train_ln_fwd.argtypes = [vp, vp, vp, vp, vp, C.c_float, u64, u64, C.c_long, vp, vp, vp, vp]
    lib.amuse_train_ln_bwd.argtypes = [vp, vp, vp, vp, vp, C.c_float, u64, u64, C.c_long, vp, vp, vp, vp, vp, vp, vp]
    lib.amuse_train_bias_gelu_drop_fwd.argtypes = [vp, vp, C.c_float, u64, u64, C.c_long, C.c_int, vp, vp]
    lib.amuse_train_bias_gelu_drop_bwd.argtypes = [vp, vp, vp, C.c_float, u64, u64, C.c_long, C.c_int, vp, vp, vp, vp]
    lib.amuse_train_colsum.argtypes = [vp, C.c_long, C.c_int, vp, vp, vp]
    lib.amuse_train_attn_fwd.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint64, vp, vp, vp, vp]
    lib.amuse_train_attn_bwd.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_uint64, C.c_uint64, vp, vp]
    lib.amuse_train_attn_fwd.restype = lib.amuse_train_attn_bwd.restype = C.c_int
    lib.amuse_train_adamw.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_long, vp]
    lib.amuse_train_adamw.restype = C.c_int
    lib.amuse_train_adamw_dev.argtypes = [vp, vp, vp, vp, C.c_size_t, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, vp, vp, C.c_int, vp]
    lib.amuse_train_epoch_advance.argtypes = [C.c_uint, vp]
    lib.amuse_train_epoch_set.argtypes = [C.c_uint, vp]
    lib.amuse_train_adamw_dev.restype = lib.amuse_train_epoch_advance.restype = lib.amuse_train_epoch_set.restype = C.c_int
    lib.amuse_train_layer_fwd.argtypes = [C.POINTER(TrainLayer), vp]
    lib.amuse_train_layer_bwd.argtypes = [C.POINTER(TrainLayer), vp]
    lib.amuse_train_linear_fwd.argtypes = [vp, vp, vp, C.c_long, C.c_int, C.c_int, vp, vp]
    lib.amuse_train_linear_bwd.argtypes = [vp, vp, vp, C.c_long, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp]
    for n in ("amuse_train_ln_fwd", "amuse_train_ln_bwd", "amuse_train_bias_gelu_drop_fwd", "amuse_train_bias_gelu_drop_bwd", "amuse_train_colsum",
              "amuse_train_layer_fwd", "amuse_train_layer_bwd", "amuse_train_linear_fwd", "amuse_train_linear_bwd"):
        getattr(lib, n).restype = C.c_int
    for n in ("amuse_audio_fbank", "amuse_audio_encode", "amuse_audio_features", "amuse_debug_gemm", "amuse_debug_tile"):
        getattr(lib, n).restype = C.c_int
    if lib.amuse_abi_version() != ABI_VERSION:
        raise AmuseHipError(f"ABI mismatch: library {lib.amuse_abi_version()} vs binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise AmuseHipError(f"libamuse_hip error {rc}: {load().amuse_last_error().decode()}")


DECODE_PATHS = ("auto", "staged", "fused", "clip")   # include/amuse_hip.h AMUSE_DECODE_*


def plan(clips_total: int, precision: int = PREC_BF16, tokens: int = 5, arch: int = ARCH_ENC) -> dict:
    """amuse_plan (include/amuse_hip.h): the kernels a job of `clips_total` clips takes - {"clips_per_group": int, "decode_path" / "encode_path" /
    "step_path": "staged" | "fused" | "clip"}.  The library's own rule, no GPU needed; nothing in Python restates it."""
    out = [C.c_int(0) for _ in range(4)]
    check(load().amuse_plan(int(arch), int(precision), int(clips_total), int(tokens), *(C.byref(o) for o in out)))
    return {"clips_per_group": out[0].value, "decode_path": DECODE_PATHS[out[1].value], "encode_path": DECODE_PATHS[out[2].value],
            "step_path": DECODE_PATHS[out[3].value]}
