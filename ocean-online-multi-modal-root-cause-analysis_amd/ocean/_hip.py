"""ctypes binding of libocean_hip.so (the C ABI declared in include/ocean_hip.h).

This module is plumbing only: it loads the shared library, declares the
signatures, converts torch tensors to device pointers and turns status codes
into exceptions.  There is no CPU fallback: every op raises if the library is
missing or a tensor is not a contiguous fp32 CUDA tensor.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_double, c_float, c_int, c_long, c_size_t, c_uint64, c_void_p
from typing import Optional, Sequence

import torch

MAX_CONV_LAYERS = 8
# OCEAN_HIP_LIB: another build of the library (A/B measurements); the default is the one built in-tree
_LIB_PATH = os.environ.get("OCEAN_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libocean_hip.so")
_lib = None

_STATUS = {1: "bad shape", 2: "unsupported size", 3: "workspace too small", 4: "null pointer",
           5: "the problem exceeds the library's built-in scratch array"}


class OceanHipError(RuntimeError):
    pass


class DcnnShape(ctypes.Structure):
    _fields_ = [("B", c_int), ("T", c_int), ("F", c_int), ("L", c_int),
                ("channels", c_int * MAX_CONV_LAYERS), ("dilations", c_int * MAX_CONV_LAYERS),
                ("kernel_size", c_int), ("out_dim", c_int), ("training", c_int),
                ("eps", c_float), ("momentum", c_float), ("bn_count", c_long),
                ("dropout", c_float), ("seed", c_uint64), ("seed_epoch", c_void_p)]


class GnnShape(ctypes.Structure):
    _fields_ = [("N", c_int), ("R", c_int), ("G", c_int), ("D", c_int), ("H", c_int), ("layers", c_int),
                ("E", c_int), ("pool_n", c_int), ("training", c_int), ("dropout", c_float),
                ("negative_slope", c_float), ("seed", c_uint64), ("seed_epoch", c_void_p)]


def _declare(lib) -> None:
    fp = POINTER(c_float)
    lib.ocean_abi_version.restype = c_int
    lib.ocean_gemm_f32.restype = c_int
    lib.ocean_gemm_f32.argtypes = [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]
    sp = POINTER(DcnnShape)
    lib.ocean_dcnn_param_count.restype = c_size_t
    lib.ocean_dcnn_param_count.argtypes = [sp]
    lib.ocean_dcnn_fwd_workspace.restype = c_size_t
    lib.ocean_dcnn_fwd_workspace.argtypes = [sp]
    lib.ocean_dcnn_bwd_workspace.restype = c_size_t
    lib.ocean_dcnn_bwd_workspace.argtypes = [sp]
    lib.ocean_dcnn_num_phases.restype = c_int
    lib.ocean_dcnn_num_phases.argtypes = [sp]
    lib.ocean_dcnn_stats_block.restype = c_int
    lib.ocean_dcnn_stats_block.argtypes = [sp, c_int, c_int, POINTER(c_size_t), POINTER(c_int)]
    lib.ocean_dcnn_fwd.restype = c_int
    lib.ocean_dcnn_fwd.argtypes = [sp, c_void_p, POINTER(c_void_p), c_void_p, c_size_t, c_void_p, c_int, c_int, c_void_p]
    lib.ocean_dcnn_bwd.restype = c_int
    lib.ocean_dcnn_bwd.argtypes = [sp, c_void_p, POINTER(c_void_p), c_void_p, c_void_p, c_size_t, c_void_p,
                                   POINTER(c_void_p), c_void_p, c_int, c_int, c_void_p]
    lib.ocean_dcnn_bwd_part.restype = c_int
    lib.ocean_dcnn_bwd_part.argtypes = lib.ocean_dcnn_bwd.argtypes


class MfaShape(ctypes.Structure):
    _fields_ = [("B", c_int), ("td", c_int), ("sd", c_int), ("ld", c_int), ("W", c_int), ("heads", c_int),
                ("training", c_int), ("dropout", c_float), ("eps", c_float), ("seed", c_uint64), ("seed_epoch", c_void_p)]


class FusionShape(ctypes.Structure):
    _fields_ = [("B", c_int), ("td", c_int), ("sd", c_int), ("ld", c_int), ("E", c_int), ("Pd", c_int),
                ("heads", c_int), ("training", c_int), ("contrastive", c_int), ("dropout", c_float),
                ("eps", c_float), ("seed", c_uint64), ("seed_epoch", c_void_p)]


class InfoNceShape(ctypes.Structure):
    _fields_ = [("B", c_int), ("Bp", c_int), ("P", c_int), ("K", c_int), ("label_offset", c_int),
                ("mean_rows", c_int), ("tau", c_float)]


class InfoNce3Shape(ctypes.Structure):
    _fields_ = [("B", c_int), ("P", c_int), ("tau", c_float), ("Bp", c_int), ("label_offset", c_int)]


class MiShape(ctypes.Structure):
    _fields_ = [("B", c_int), ("E", c_int)]


class HeadShape(ctypes.Structure):
    _fields_ = [("B", c_int), ("E", c_int), ("h1", c_int), ("h2", c_int), ("S", c_int), ("training", c_int),
                ("dropout", c_float), ("seed", c_uint64), ("seed_epoch", c_void_p)]


class ProjShape(ctypes.Structure):
    _fields_ = [("B", c_int), ("in_dim", c_int), ("hidden", c_int), ("out_dim", c_int), ("training", c_int),
                ("normalize", c_int), ("dropout", c_float), ("seed", c_uint64), ("seed_epoch", c_void_p)]


class MhaShape(ctypes.Structure):
    _fields_ = [("B", c_int), ("L", c_int), ("D", c_int), ("heads", c_int), ("training", c_int), ("project", c_int),
                ("dropout", c_float), ("scale", c_float), ("seed", c_uint64), ("seed_epoch", c_void_p)]


class XattnTailShape(ctypes.Structure):
    _fields_ = [("rows", c_int), ("D", c_int), ("F", c_int), ("training", c_int), ("dropout", c_float), ("eps", c_float),
                ("seed", c_uint64), ("seed_epoch", c_void_p)]


class ConvBlockShape(ctypes.Structure):
    _fields_ = [("B", c_int), ("T", c_int), ("Cin", c_int), ("Cout", c_int), ("kernel", c_int), ("dilation", c_int),
                ("training", c_int), ("res_conv", c_int), ("dropout", c_float), ("eps", c_float), ("momentum", c_float),
                ("seed", c_uint64), ("seed_epoch", c_void_p)]


class GatConvShape(ctypes.Structure):
    _fields_ = [("N", c_int), ("in_dim", c_int), ("H", c_int), ("C", c_int), ("E", c_int), ("concat", c_int), ("training", c_int),
                ("negative_slope", c_float), ("dropout", c_float), ("seed", c_uint64), ("seed_epoch", c_void_p)]


class AdamChunk(ctypes.Structure):
    _fields_ = [("param", c_uint64), ("grad", c_uint64), ("exp_avg", c_uint64), ("exp_avg_sq", c_uint64),
                ("n", ctypes.c_int64)]


def _declare_rest(lib) -> None:
    vp, tp = c_void_p, POINTER(c_void_p)

    def sig(name, res, *args):
        f = getattr(lib, name)
        f.restype, f.argtypes = res, list(args)

    mp, fp_, ip, hp = POINTER(MfaShape), POINTER(FusionShape), POINTER(InfoNceShape), POINTER(HeadShape)
    sig("ocean_mfa_fwd_workspace", c_size_t, mp)
    sig("ocean_mfa_bwd_workspace", c_size_t, mp)
    sig("ocean_mfa_fwd", c_int, mp, vp, vp, vp, tp, vp, c_size_t, vp, vp)
    sig("ocean_mfa_bwd", c_int, mp, vp, vp, vp, tp, vp, vp, c_size_t, vp, tp, vp, vp, vp, vp)
    sig("ocean_fusion_fwd_workspace", c_size_t, fp_)
    sig("ocean_fusion_bwd_workspace", c_size_t, fp_)
    sig("ocean_fusion_fwd", c_int, fp_, vp, vp, vp, tp, vp, c_size_t, vp, vp, vp, vp)
    sig("ocean_fusion_bwd", c_int, fp_, vp, vp, vp, tp, vp, vp, vp, vp, c_size_t, vp, vp, vp, tp, vp, vp, vp, vp)
    sig("ocean_infonce_workspace", c_size_t, ip)
    sig("ocean_infonce_bwd_workspace", c_size_t, ip)
    sig("ocean_infonce_fwd", c_int, ip, vp, vp, vp, vp, c_size_t, vp, vp)
    sig("ocean_infonce_bwd", c_int, ip, vp, vp, vp, vp, c_size_t, vp, vp, vp, vp)
    i3 = ctypes.POINTER(InfoNce3Shape)
    sig("ocean_infonce3_workspace", c_size_t, i3)
    sig("ocean_infonce3_bwd_workspace", c_size_t, i3)
    sig("ocean_infonce3_fwd", c_int, i3, vp, vp, vp, c_size_t, vp, vp)
    sig("ocean_infonce3_bwd", c_int, i3, vp, vp, c_size_t, vp, vp, vp, vp)
    mp = ctypes.POINTER(MiShape)
    sig("ocean_mi_workspace", c_size_t, mp)
    sig("ocean_mi_estimates", c_int, mp, vp, vp, vp, vp, vp, vp, vp, vp, c_size_t, vp, vp)
    sig("ocean_head_workspace", c_size_t, hp)
    sig("ocean_head_bwd_workspace", c_size_t, hp)
    sig("ocean_head_fwd", c_int, hp, vp, tp, vp, c_size_t, vp, vp, vp)
    sig("ocean_head_bwd", c_int, hp, vp, tp, vp, vp, c_size_t, vp, vp, vp, tp, vp, vp)
    pp = POINTER(ProjShape)
    sig("ocean_projection_head_fwd_workspace", c_size_t, pp)
    sig("ocean_projection_head_bwd_workspace", c_size_t, pp)
    sig("ocean_projection_head_fwd", c_int, pp, vp, tp, vp, c_size_t, vp, vp)
    sig("ocean_projection_head_bwd", c_int, pp, vp, tp, vp, vp, vp, vp, c_size_t, tp, vp, vp)
    ap_ = POINTER(MhaShape)
    sig("ocean_mha_fwd_workspace", c_size_t, ap_)
    sig("ocean_mha_bwd_workspace", c_size_t, ap_)
    sig("ocean_mha_fwd", c_int, ap_, vp, vp, vp, vp, tp, vp, c_size_t, vp, vp, vp)
    sig("ocean_mha_bwd", c_int, ap_, vp, vp, vp, tp, vp, vp, vp, vp, vp, c_size_t, tp, vp, vp, vp, vp)
    xp = POINTER(XattnTailShape)
    sig("ocean_xattn_tail_fwd_workspace", c_size_t, xp)
    sig("ocean_xattn_tail_bwd_workspace", c_size_t, xp)
    sig("ocean_xattn_tail_fwd", c_int, xp, vp, vp, tp, vp, c_size_t, vp, vp)
    sig("ocean_xattn_tail_bwd", c_int, xp, tp, vp, vp, vp, c_size_t, tp, vp, vp)
    cp_ = POINTER(ConvBlockShape)
    sig("ocean_convblock_fwd_workspace", c_size_t, cp_)
    sig("ocean_convblock_bwd_workspace", c_size_t, cp_)
    sig("ocean_convblock_fwd", c_int, cp_, vp, tp, vp, c_size_t, vp, vp)
    sig("ocean_convblock_bwd", c_int, cp_, vp, tp, vp, vp, vp, c_size_t, tp, vp, vp)
    gp_ = POINTER(GatConvShape)
    sig("ocean_gatconv_fwd_workspace", c_size_t, gp_)
    sig("ocean_gatconv_bwd_workspace", c_size_t, gp_)
    sig("ocean_gatconv_fwd", c_int, gp_, vp, tp, tp, vp, c_size_t, vp, vp)
    sig("ocean_gatconv_bwd", c_int, gp_, vp, tp, tp, vp, vp, vp, c_size_t, tp, vp, vp)
    sig("ocean_bce_fwd", c_int, vp, vp, c_long, c_long, vp, vp)
    sig("ocean_bce_bwd", c_int, vp, vp, vp, c_long, c_long, vp, vp)
    sig("ocean_conv_gemm_f32", c_int, vp, c_int, vp, c_int, vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_long, c_long, c_int, vp)
    sig("ocean_linear_bwd_f32", c_int, vp, c_int, vp, c_int, vp, vp, vp, vp, c_int, c_int, c_int, c_int, c_int, vp)
    sig("ocean_mean_mid_f32", c_int, vp, vp, c_int, c_int, c_int, vp)
    sig("ocean_draw_permutations", c_int, c_int, c_int, c_uint64, vp, vp, vp)
    sig("ocean_loss_combine_fwd", c_int, vp, vp, c_int, c_float, c_float, vp, vp)
    sig("ocean_loss_combine_bwd", c_int, vp, vp, c_int, c_float, c_float, vp, vp, vp)
    sig("ocean_pred_total_fwd", c_int, vp, vp, c_long, c_long, vp, c_int, c_float, c_float, vp, vp, vp)
    sig("ocean_pred_total_bwd", c_int, vp, vp, vp, vp, vp, c_long, c_long, c_int, c_float, c_float, vp, vp, vp)
    sig("ocean_mse_fwd", c_int, vp, vp, c_long, vp, vp)
    sig("ocean_mse_bwd", c_int, vp, vp, vp, c_long, vp, vp)
    sig("ocean_diversity_workspace", c_size_t, c_int, c_int)
    sig("ocean_diversity_fwd", c_int, vp, c_int, c_int, vp, c_size_t, vp, vp)
    sig("ocean_diversity_bwd", c_int, c_int, c_int, vp, vp, vp, vp)
    sig("ocean_bank_ring_write", c_int, vp, vp, vp, c_int, c_int, c_int, vp)
    sig("ocean_infonce_fwd_static", c_int, ip, vp, vp, vp, vp, vp)
    sig("ocean_static_scratch_warmup", c_int)
    sig("ocean_rc_debug_trace", None, vp)
    sig("ocean_infonce_rows_fwd", c_int, ip, vp, vp, vp, vp, c_size_t, vp, vp)
    sig("ocean_infonce_rows_bwd", c_int, ip, vp, vp, vp, vp, c_size_t, vp, vp, vp, vp)
    sig("ocean_bce_elem_fwd", c_int, vp, vp, c_long, vp, vp)
    sig("ocean_bce_elem_bwd", c_int, vp, vp, vp, c_long, vp, vp)
    sig("ocean_dcnn_block_output", c_int, POINTER(DcnnShape), c_int, POINTER(c_size_t), POINTER(c_int))
    sig("ocean_dcnn_pooled_output", c_int, POINTER(DcnnShape), POINTER(c_size_t), POINTER(c_int))
    sig("ocean_gnn_layer_output", c_int, POINTER(GnnShape), c_int, POINTER(c_size_t), POINTER(c_int))
    sig("ocean_gnn_edge_kernel_family", c_int, POINTER(GnnShape))
    sig("ocean_prof_enable", c_int, c_int)
    sig("ocean_set_split_bf16", c_int, c_int)
    sig("ocean_get_split_bf16", c_int)
    sig("ocean_clear_last_error", c_int)
    sig("ocean_prof_summary", c_int, ctypes.c_char_p, c_size_t)
    sig("ocean_grad_sqnorm", c_int, vp, c_int, vp, vp)
    sig("ocean_clip_adam_step", c_int, vp, c_int, vp, c_float, c_float, c_float, c_float, c_float, c_float, c_int, c_int, vp)
    sig("ocean_clip_adam_step_dev", c_int, vp, c_int, vp, c_float, vp, vp, c_float, c_float, c_float, c_float, c_int, vp)
    sig("ocean_clip_adam_step_auto", c_int, vp, c_int, vp, c_float, vp, vp, c_float, c_float, c_float, c_float, c_int, vp)


def _declare_gnn(lib) -> None:
    gp = POINTER(GnnShape)
    lib.ocean_gnn_fwd_workspace.restype = c_size_t
    lib.ocean_gnn_fwd_workspace.argtypes = [gp]
    lib.ocean_gnn_bwd_workspace.restype = c_size_t
    lib.ocean_gnn_bwd_workspace.argtypes = [gp]
    lib.ocean_gnn_fwd.restype = c_int
    lib.ocean_gnn_fwd.argtypes = [gp, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_size_t, c_void_p, c_void_p]
    lib.ocean_gnn_bwd.restype = c_int
    lib.ocean_gnn_bwd.argtypes = [gp, c_void_p, POINTER(c_void_p), POINTER(c_void_p), c_void_p, c_void_p, c_size_t,
                                  c_void_p, POINTER(c_void_p), c_void_p]


def lib():
    """The loaded library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise OceanHipError(
                f"{_LIB_PATH} is missing: build it with `make -C <package>/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        _lib = ctypes.CDLL(_LIB_PATH)
        _declare(_lib)
        _declare_gnn(_lib)
        _declare_rest(_lib)
    return _lib


def check(status: int, what: str) -> None:
    if status == 0:
        return
    if status >= 1000:
        raise OceanHipError(f"{what}: HIP error {status - 1000}")
    raise OceanHipError(f"{what}: {_STATUS.get(status, 'status ' + str(status))}")


_SPLIT_MODES = {False: 0, True: 1, 0: 0, 1: 1, 2: 2, "off": 0, "all": 1, "backward": 2}


def split_bf16() -> int:
    """The arithmetic of the large convolution products: 0 fp32, 1 split bf16 everywhere, 2 split bf16 in the backward
    pass only (see set_split_bf16)."""
    return int(lib().ocean_get_split_bf16())


def set_split_bf16(mode) -> int:
    """Opt-in arithmetic for the large (full-tile) products of the convolutions and their gradients: every fp32 operand
    is split into two bf16 values and the product formed by three bf16 matrix instructions with fp32 accumulation -
    about 16 bits of each product instead of 24, at 2-2.5 times the speed.  `mode`: False / "off"; True / "all" (forward
    and backward); "backward" (the two gradient products only: the forward pass, and so every ReLU's sign, stays fp32).
    Process-wide; returns the previous mode (0, 1 or 2).  Recorded HIP graphs are keyed on it, so a learner records a new
    step after a change."""
    if mode not in _SPLIT_MODES:
        raise ValueError(f"split-bf16 mode must be one of False, True, 'off', 'all', 'backward', 0, 1, 2 (got {mode!r})")
    return check_mode(lib().ocean_set_split_bf16(_SPLIT_MODES[mode]))


def check_mode(rc: int) -> int:
    if rc < 0:
        raise OceanHipError(f"ocean_set_split_bf16 status {rc}")
    return int(rc)


def dev_ptr(t: Optional[torch.Tensor], name: str = "tensor", dtype=torch.float32) -> c_void_p:
    if t is None:
        return c_void_p(0)
    if not t.is_cuda:
        raise OceanHipError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != dtype:
        raise OceanHipError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise OceanHipError(f"{name} must be contiguous")
    return c_void_p(t.data_ptr())


def ptr_table(tensors: Sequence[Optional[torch.Tensor]], name: str):
    arr = (c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = dev_ptr(t, f"{name}[{i}]").value if t is not None else None
    return arr


_epochs = {}


def seed_epoch(device) -> torch.Tensor:
    """The per-device dropout epoch counter (int64[1], starts at 0).  Every kernel that draws a dropout mask adds
    it into its seed on the device, so a captured HIP graph - whose kernel arguments, seeds included, are frozen -
    still draws new masks on each replay once the replay increments the counter."""
    device = torch.device(device)
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    t = _epochs.get(key)
    if t is None:
        t = torch.zeros(1, dtype=torch.int64, device=device)
        _epochs[key] = t
    return t


def new_seed(p: float) -> int:
    """A fresh dropout seed from torch's CPU generator (0 when dropout is inactive)."""
    return int(torch.randint(0, 2 ** 62, (1,)).item()) if p > 0 else 0


class GradArena:
    """One persistent flat fp32 buffer holding the gradient of every parameter of a model, in parameter order (each slot
    padded to 16 bytes).  While an arena is registered, the component drivers write their parameter gradients into its
    slots (zero_grads_like below) instead of into a fresh allocation per backward pass, so that after backward every
    p.grad is a view of the arena: a data-parallel caller all-reduces the arena in place (no packing into a bucket and
    back), and the optimizer's chunk table, which holds gradient addresses, never changes.  Assumes each component
    runs once per backward pass (a second use would zero the first one's gradients instead of adding to them).

    step_scoped: the arena serves gradient buffers only between begin_step() and end_step() (an update step of a learner,
    which calls both); any other backward pass through the same parameters gets fresh allocations, so callers that
    accumulate gradients over several backward passes are not affected.  begin_step() zeroes the whole arena with one
    fill, and the components' requests inside the step then find their slots already zero."""

    def __init__(self, params, step_scoped: bool = False):
        params = [p for p in params]
        if not params:
            raise OceanHipError("GradArena needs parameters")
        self.device = params[0].device
        self.slots = {}
        off = 0
        for p in params:
            if p.device != self.device or p.dtype != torch.float32:
                raise OceanHipError("GradArena: parameters must be fp32 on one device")
            self.slots[p.data_ptr()] = (off, p.numel(), tuple(p.shape))
            off += (p.numel() + 3) // 4 * 4
        self.flat = torch.zeros(off, device=self.device, dtype=torch.float32)
        self.step_scoped = bool(step_scoped)
        self._in_step = False

    def begin_step(self) -> None:
        """One fill for all gradients of the coming backward pass."""
        self.flat.zero_()
        self._in_step = True

    def end_step(self) -> None:
        self._in_step = False

    def offset_of(self, p: torch.Tensor) -> int:
        return self.slots[p.data_ptr()][0]

    def take(self, params, skip=None):
        """Views of the slots of `params` after zeroing the span they cover (one fill), or None when a parameter is
        not in this arena."""
        if self.step_scoped and not self._in_step:
            return None
        lo, hi, found = None, 0, []
        for i, p in enumerate(params):
            if skip is not None and skip[i]:
                found.append(None)
                continue
            slot = self.slots.get(p.data_ptr())
            if slot is None or slot[2] != tuple(p.shape):
                return None
            found.append(slot)
            lo = slot[0] if lo is None else min(lo, slot[0])
            hi = max(hi, slot[0] + slot[1])
        if lo is None:
            return [None] * len(params)
        if not self._in_step:
            self.flat[lo:hi].zero_()
        return [None if sl is None else self.flat[sl[0]: sl[0] + sl[1]].view(sl[2]) for sl in found]


_arenas: list = []


def register_grad_arena(arena: Optional[GradArena]) -> None:
    if arena is not None and all(a is not arena for a in _arenas):
        _arenas.append(arena)


def unregister_grad_arena(arena: GradArena) -> None:
    _arenas[:] = [a for a in _arenas if a is not arena]


def step_arena(owner) -> Optional[GradArena]:
    """The arena an update step of `owner` (an OnlineLearner or a Trainer: .model, .optimizer, .data_parallel, .grad_arena,
    ._arena) writes its parameter gradients into: the data-parallel wiring's when there is one, else the owner's own
    step-scoped one (made on first use for the model's current parameter tensors), or None (switched off, not on a GPU,
    not the fused optimizer)."""
    dp_arena = getattr(getattr(owner, "data_parallel", None), "arena", None)
    if dp_arena is not None:
        return dp_arena
    model = owner.model
    from .optim import FusedAdam
    last = model.prediction_head[0].weight
    if not getattr(owner, "grad_arena", False) or last.device.type != "cuda" or not isinstance(owner.optimizer, FusedAdam):
        return None
    a = owner._arena
    # two parameters from either end of the model stand for all of them: a move to another device or a reload that
    # replaces tensors changes both (walking all 201 parameters on every step costs more host time than the fills the
    # arena saves).  A component whose tensors are not in the arena simply gets fresh buffers (GradArena.take).
    first = model.temporal_encoder.input_projection.weight
    if a is None or a.device != last.device or first.data_ptr() not in a.slots or last.data_ptr() not in a.slots:
        if a is not None:
            unregister_grad_arena(a)
        a = owner._arena = GradArena([p for p in model.parameters() if p.dtype == torch.float32], step_scoped=True)
        register_grad_arena(a)
    return a


def zero_grads_like(params, skip=None):
    """Zero-initialised gradient buffers for `params` as views of ONE flat allocation (one fill kernel
    instead of one per tensor).  skip[i] True -> None for that slot.  With a registered GradArena that holds these
    parameters the buffers are its slots."""
    for arena in _arenas:
        got = arena.take(params, skip)
        if got is not None:
            return got
    sizes = [0 if (skip is not None and skip[i]) else (p.numel() + 3) // 4 * 4 for i, p in enumerate(params)]
    total = sum(sizes)
    if total == 0:
        return [None] * len(params)
    flat = torch.zeros(total, device=params[0].device, dtype=torch.float32)
    out, off = [], 0
    for p, n in zip(params, sizes):
        out.append(None if n == 0 else flat[off: off + p.numel()].view_as(p))
        off += n
    return out


def capture(graph: "torch.cuda.CUDAGraph"):
    """torch.cuda.graph(graph) in thread-local capture mode.  In the default (global) mode the HIP runtime refuses
    potentially unsafe calls from EVERY thread while the stream is capturing; the RCCL process group's watchdog thread
    queries the events of earlier collectives on its own schedule, gets hipErrorStreamCaptureUnsupported when that falls
    into a capture, and takes the process down (seen about once in ten recordings of a data-parallel step).  Other
    threads of an application - a loader pinning memory, a copy - are in the same position.  The recorded work itself
    is launched from this thread and from the autograd engine's worker onto the capturing stream and is captured
    either way; the mode only decides whose unrelated calls are refused."""
    return torch.cuda.graph(graph, capture_error_mode="thread_local")


def stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def gemm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, M: int, N: int, K: int, *, lda=None, ldb=None,
         ldc=None, a_kmajor=False, b_kn=False, bias=None, residual=None, ldr=0, relu=False) -> torch.Tensor:
    """Raw GEMM entry (tests and small glue): see ocean_gemm_f32."""
    lda = lda if lda is not None else (M if a_kmajor else K)
    ldb = ldb if ldb is not None else (N if b_kn else K)
    ldc = ldc if ldc is not None else N
    st = lib().ocean_gemm_f32(dev_ptr(a, "A"), lda, dev_ptr(b, "B"), ldb, dev_ptr(out, "C"), ldc, M, N, K,
                              int(a_kmajor), int(b_kn), dev_ptr(bias, "bias"), dev_ptr(residual, "R"),
                              ldr or ldc, int(relu), stream())
    check(st, "ocean_gemm_f32")
    return out


def linear_bwd(x: torch.Tensor, dy: torch.Tensor, w: torch.Tensor, dw: torch.Tensor, db, dx: torch.Tensor, M: int,
               N: int, K: int, *, ldx=None, lddy=None, lddx=None, accumulate_dx=False) -> None:
    """Both gradients of one linear layer (tests and small glue): see ocean_linear_bwd_f32."""
    st = lib().ocean_linear_bwd_f32(dev_ptr(x, "x"), ldx or K, dev_ptr(dy, "dy"), lddy or N, dev_ptr(w, "W"), dev_ptr(dw, "dW"),
                                    dev_ptr(db, "db"), dev_ptr(dx, "dx"), lddx or K, M, N, K, int(accumulate_dx), stream())
    check(st, "ocean_linear_bwd_f32")


def clear_last_error() -> int:
    """Drop an error the HIP runtime still holds from a failed operation of the caller (e.g. an invalidated stream
    capture), so that the next library call does not report it as its own."""
    return int(lib().ocean_clear_last_error())


def profile_enable(on: bool) -> None:
    check(lib().ocean_prof_enable(int(on)), "ocean_prof_enable")


def profile_summary():
    """Per-kernel totals of the GEMM launches recorded since profile_enable(True)."""
    import json
    buf = ctypes.create_string_buffer(1 << 16)
    check(lib().ocean_prof_summary(buf, len(buf)), "ocean_prof_summary")
    return json.loads(buf.value.decode())
