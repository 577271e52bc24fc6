"""Multiresolution grid encoder module — same constructor, attributes, forward() and grad_total_variation() as
the reference's `gridencoder/grid.py` (cited per item), backed by libcustomnerf_hip.so.

Deliberate differences:
  * the kernel-layout output [L,B,C] is also reachable (`forward(..., return_kernel_layout=True)`) so the fused field
    kernel can consume it without the permute+reshape copy of grid.py:63;
  * under autocast the fp16 table is a cached shadow copy refreshed only when the fp32 parameter changes (the
    reference re-casts the whole table on every call, grid.py:45-46); the change is detected through the parameter's version
    counter, so writes through `.data` must be followed by invalidate_half_table();
  * gradients are accumulated in float32 for both table dtypes (the reference scatters __half2 atomics for fp16).
"""
import os
import weakref

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function

from .._lib import lib, check, ptr, stream, dtype_id, require_cuda, scratch_key, grad_chain_wait, grad_chain_record, scratch_reallocated

# Optional in-stream timing of the gather kernel (bench.py's roofline leg): when a list is installed with
# set_profile(), every forward launch is bracketed by a pair of events recorded on the launch stream.
_PROFILE = None


def set_profile(sink):
    """sink: None (off) or a list that receives (start_event, end_event, n_points, n_levels, itemsize) tuples."""
    global _PROFILE
    _PROFILE = sink


def forward_kernel_name(half):
    """name of the gather kernel the CustomNeRF configuration launches (bench.py's roofline record)"""
    return "k_grid_fwd_fast" if half else "k_grid_fwd"


LEVEL_MAJOR, SAMPLE_MAJOR = 0, 1          # include/customnerf_hip.h CNERF_GRID_LEVEL_MAJOR / CNERF_GRID_SAMPLE_MAJOR


class TraversalTuner:
    """Which traversal of the forward gather (cnerf_grid_encode_forward_ordered) a call site should use, MEASURED in place: the two forms are
    bit-identical, and which one is faster depends on the scene's state — importance samples of a random-initialised field are spread over the
    volume (level-major wins), those of a fitted field hug a surface (sample-major wins: profiles/r05_fused_fwd_lab.log).  On a trial call the
    caller runs BOTH forms back to back on the same samples (the second overwrites the first's identical output) between event pairs; the
    events are read without blocking on a later call, and the faster form is used until the next trial.  Two trial calls (order swapped) at
    calls `first` / `first + 1`, then every `period` calls: 2 extra gathers per `period` steps."""

    def __init__(self, first=2, period=256, margin=0.02):
        self.choice = LEVEL_MAJOR
        self.calls = 0
        self.first, self.period, self.margin = first, period, margin
        self.pending = []                 # [(traversal, e0, e1)]
        self.history = []                 # [(call index, ms level-major, ms sample-major)]: diagnostics (bench.py prints the last entry)
        self.enabled = True

    def plan(self):
        """-> list of traversals to launch for this call (one entry normally, both on a trial call: the LAST one's output stays)"""
        i = self.calls
        self.calls += 1
        if not self.enabled or torch.cuda.is_current_stream_capturing():      # (no event may be recorded or queried while a hipGraph is captured)
            return [self.choice]
        self._harvest()
        k = (i - self.first) % self.period if i >= self.first else -1
        if k == 0:
            return [LEVEL_MAJOR, SAMPLE_MAJOR]
        if k == 1:
            return [SAMPLE_MAJOR, LEVEL_MAJOR]
        return [self.choice]

    def record(self, traversal, e0, e1):
        self.pending.append((traversal, e0, e1))

    def _harvest(self):
        if len(self.pending) < 4 or not all(e1.query() for _, _, e1 in self.pending):
            return
        ms = [0.0, 0.0]
        for t, e0, e1 in self.pending:
            ms[t] += e0.elapsed_time(e1)
        self.pending = []
        self.history.append((self.calls, ms[0] / 2, ms[1] / 2))
        if ms[SAMPLE_MAJOR] < ms[LEVEL_MAJOR] * (1 - self.margin):
            self.choice = SAMPLE_MAJOR
        elif ms[LEVEL_MAJOR] < ms[SAMPLE_MAJOR] * (1 - self.margin):
            self.choice = LEVEL_MAJOR


_gridtype_to_id = {'hash': 0, 'tiled': 1}
_interp_to_id = {'linear': 0, 'smoothstep': 1}


def _offsets_host(offsets):
    """int32 numpy copy of the level offset table (the C-ABI takes it as a host array)."""
    if isinstance(offsets, np.ndarray):
        return np.ascontiguousarray(offsets, dtype=np.int32)
    cached = getattr(offsets, "_cnerf_host", None)
    if cached is None:
        cached = np.ascontiguousarray(offsets.detach().cpu().numpy(), dtype=np.int32)
        try:
            offsets._cnerf_host = cached
        except Exception:
            pass
    return cached


_WS_CACHE = {}


def _bwd_workspace(offsets_host, B, D, C, L, max_level, S, H, dt, device):
    """Scratch for the atomic-free binned scatter (one buffer per device, grown on demand, reused every step)."""
    import ctypes
    need = ctypes.c_uint64(0)
    check(lib.cnerf_grid_encode_backward_workspace_bytes(offsets_host.ctypes.data, B, D, C, L, max_level, S, H, dt, ctypes.addressof(need)),
          "grid_encode_backward_workspace_bytes")
    if need.value == 0:
        return None, 0
    key = scratch_key(device)
    buf = _WS_CACHE.get(key)
    if buf is None or buf.numel() < need.value:
        buf = torch.empty(int(need.value * 1.25) + 256, dtype=torch.uint8, device=device)
        _WS_CACHE[key] = buf
        scratch_reallocated()
    return buf, buf.numel()


class _grid_encode(Function):
    """grid.py:24-95.  `embeddings` is the float32 parameter; `table` is what the kernel reads (the parameter itself, or
    its fp16 shadow).  Returns the encoder output in kernel layout [L,B,C]."""

    @staticmethod
    def forward(ctx, inputs, embeddings, table, offsets_host, per_level_scale, base_resolution, calc_grad_inputs, gridtype,
                align_corners, interpolation, max_level):
        require_cuda(inputs, table)
        inputs = inputs.contiguous().float()
        B, D = inputs.shape
        L = offsets_host.shape[0] - 1
        C = table.shape[1]
        S = float(np.log2(per_level_scale))
        H = int(base_resolution)
        max_level = L if max_level is None else min(max_level, L)
        outputs = torch.empty(L, B, C, device=inputs.device, dtype=table.dtype)
        if max_level < L:
            outputs.zero_()
        if calc_grad_inputs:
            dy_dx = torch.empty(B, L * D * C, device=inputs.device, dtype=table.dtype)
            if max_level < L:
                dy_dx.zero_()
        else:
            dy_dx = None
        prof = _PROFILE
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        check(lib.cnerf_grid_encode_forward(ptr(inputs), ptr(table), offsets_host.ctypes.data, ptr(outputs), B, D, C, L, max_level, S, H,
                                            ptr(dy_dx), gridtype, int(align_corners), interpolation, dtype_id(table), stream()),
              "grid_encode_forward")
        if prof is not None:
            e1.record()
            prof.append((e0, e1, B, max_level, table.element_size()))
        ctx.save_for_backward(inputs, dy_dx)
        ctx.cfg = (offsets_host, B, D, C, L, S, H, gridtype, interpolation, max_level, align_corners, tuple(embeddings.shape))
        return outputs

    @staticmethod
    def backward(ctx, grad):
        inputs, dy_dx = ctx.saved_tensors
        offsets_host, B, D, C, L, S, H, gridtype, interpolation, max_level, align_corners, eshape = ctx.cfg
        grad = grad.contiguous()                                     # already [L,B,C]: no permute copy (grid.py:81)
        if dy_dx is not None and grad.dtype != dy_dx.dtype:
            grad = grad.to(dy_dx.dtype)
        grad_embeddings = torch.zeros(eshape, device=grad.device, dtype=torch.float32)
        grad_inputs = torch.empty(B, D, device=grad.device, dtype=torch.float32) if dy_dx is not None else None
        ws, ws_bytes = (None, 0) if dy_dx is not None else _bwd_workspace(offsets_host, B, D, C, L, max_level, S, H, dtype_id(grad), grad.device)
        check(lib.cnerf_grid_encode_backward(ptr(grad), ptr(inputs), offsets_host.ctypes.data, ptr(grad_embeddings), B, D, C, L, max_level,
                                             S, H, ptr(dy_dx), ptr(grad_inputs), gridtype, int(align_corners), interpolation,
                                             dtype_id(grad), ptr(ws), ws_bytes, stream()), "grid_encode_backward")
        return grad_inputs, grad_embeddings, None, None, None, None, None, None, None, None, None


_PRE_SCATTER_HOOK = None


def set_pre_scatter_hook(fn):
    """fn() is called at the top of the grid scatter's backward, i.e. after the field backward has been enqueued and before the scatter
    kernels are: the data-parallel layer starts the MLP gradients' all-reduce there so that it overlaps the scatter (customnerf_amd.dp)."""
    global _PRE_SCATTER_HOOK
    _PRE_SCATTER_HOOK = fn


_SIDE = {}          # device -> {'stream': side stream, 'ws': workspace of the prepared plan, 'owner': weakref to the plan waiting for its backward}


class _Plan:
    """a prepared scatter plan (lives on the autograd node: dropped with the graph if the backward never runs)"""
    __slots__ = ('ws', 'event', 'inputs_ptr', 'rows', '__weakref__')

    def __init__(self, ws, event):
        self.ws, self.event = ws, event


def _side(device):
    key = scratch_key(device)                                # one side stream + plan slot per compute stream (micro-batch pipeline)
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = {'stream': torch.cuda.Stream(device=device), 'ws': None, 'owner': None}
    return st


def _side_busy(side):
    return side['owner'] is not None and side['owner']() is not None


_NEEDS_PLAN = {}


def _needs_plan(offsets_host, B, D, C, L, S, H, dt, gridtype):
    """does the backward scatter of this shape use a plan prepared ahead of time?  (cnerf_grid_encode_backward_needs_plan, cached per shape: the
    round-5 scatter counts inside its emit kernel, so the run() path of the benchmark configuration prepares nothing)"""
    import ctypes
    key = (offsets_host.ctypes.data, B, D, C, L, S, H, dt, gridtype)
    v = _NEEDS_PLAN.get(key)
    if v is None:
        n = ctypes.c_int(0)
        check(lib.cnerf_grid_encode_backward_needs_plan(offsets_host.ctypes.data, B, D, C, L, L, S, H, gridtype, dt, ctypes.addressof(n)),
              "grid_encode_backward_needs_plan")
        if len(_NEEDS_PLAN) > 256:
            _NEEDS_PLAN.clear()
        v = _NEEDS_PLAN[key] = (bool(n.value), offsets_host)             # (the array reference pins the key's address)
    return v[0]


def _prepare_plan(inputs, offsets_host, B, D, C, L, S, H, gridtype, align_corners, interpolation, dt, device):
    """Issue the coordinate-only half of the binned backward scatter (histogram + scans) for `inputs` [B, D] on the side stream of the
    current compute stream -> _Plan, or None when the side stream still holds an earlier plan / the problem takes the atomic kernel."""
    import ctypes
    if not _needs_plan(offsets_host, B, D, C, L, S, H, dt, gridtype):
        return None
    side = _side(device)
    if _side_busy(side):
        return None
    need = ctypes.c_uint64(0)
    check(lib.cnerf_grid_encode_backward_workspace_bytes(offsets_host.ctypes.data, B, D, C, L, L, S, H, dt, ctypes.addressof(need)),
          "grid_encode_backward_workspace_bytes")
    if not need.value:
        return None
    if side['ws'] is None or side['ws'].numel() < need.value:
        side['ws'] = torch.empty(int(need.value * 1.25) + 256, dtype=torch.uint8, device=device)
        scratch_reallocated()
    ws = side['ws']
    cur = torch.cuda.current_stream()
    side['stream'].wait_stream(cur)                              # the coordinates were produced on the current stream
    ok = ctypes.c_int(0)
    check(lib.cnerf_grid_encode_backward_prepare(ptr(inputs), offsets_host.ctypes.data, B, D, C, L, L, S, H, gridtype, int(align_corners),
                                                 interpolation, dt, ptr(ws), ws.numel(), ctypes.addressof(ok), side['stream'].cuda_stream),
          "grid_encode_backward_prepare")
    if not ok.value:
        return None
    ev = torch.cuda.Event()
    ev.record(side['stream'])
    plan = _Plan(ws, ev)
    plan.inputs_ptr, plan.rows = inputs.data_ptr(), B
    side['owner'] = weakref.ref(plan)
    return plan


def _plan_rows(state, inputs, offsets_host, B, D, C, L, S, H, gridtype, align_corners, interpolation, dt, device, row0, rows, finish):
    """The plan of _prepare_plan in pieces (cnerf_grid_encode_backward_prepare_rows / _finish): `state` is None for the first piece and the
    value returned by the previous call afterwards; every piece is issued on the side stream after the current stream's work so far (the
    coordinates of ITS rows must exist).  -> state (a dict) while pieces are pending, a _Plan once `finish`, or None when the shape takes the
    atomic kernel / the side stream is busy (then nothing was issued and attach_backward plans by itself)."""
    import ctypes
    if state is None and not _needs_plan(offsets_host, B, D, C, L, S, H, dt, gridtype):
        return None
    side = _side(device)
    if state is None:
        if _side_busy(side):
            return None
        need = ctypes.c_uint64(0)
        check(lib.cnerf_grid_encode_backward_workspace_bytes(offsets_host.ctypes.data, B, D, C, L, L, S, H, dt, ctypes.addressof(need)),
              "grid_encode_backward_workspace_bytes")
        if not need.value:
            return None
        if side['ws'] is None or side['ws'].numel() < need.value:
            side['ws'] = torch.empty(int(need.value * 1.25) + 256, dtype=torch.uint8, device=device)
            scratch_reallocated()
        state = {'ws': side['ws'], 'token': _Plan(side['ws'], None)}
        state['token'].inputs_ptr, state['token'].rows = inputs.data_ptr(), B
        side['owner'] = weakref.ref(state['token'])                  # the side stream's workspace is taken from the first piece on
    ws = state['ws']
    cur = torch.cuda.current_stream()
    side['stream'].wait_stream(cur)
    ok = ctypes.c_int(0)
    if rows:
        check(lib.cnerf_grid_encode_backward_prepare_rows(ptr(inputs), offsets_host.ctypes.data, B, D, C, L, S, H, gridtype, int(align_corners),
                                                          interpolation, dt, int(row0), int(rows), ptr(ws), ws.numel(), ctypes.addressof(ok),
                                                          side['stream'].cuda_stream), "grid_encode_backward_prepare_rows")
        if not ok.value:
            side['owner'] = None
            return None
    if not finish:
        return state
    check(lib.cnerf_grid_encode_backward_prepare_finish(offsets_host.ctypes.data, B, D, C, L, S, H, gridtype, interpolation, dt, ptr(ws), ws.numel(),
                                                        ctypes.addressof(ok), side['stream'].cuda_stream), "grid_encode_backward_prepare_finish")
    if not ok.value:
        side['owner'] = None
        return None
    plan = state['token']
    plan.event = torch.cuda.Event()
    plan.event.record(side['stream'])
    return plan


class _grid_attach(Function):
    """Backward half of _grid_encode for a feature buffer that was filled by GridEncoder.encode_into calls: forward hands the buffer
    on unchanged, backward scatters d(loss)/d(features) [L,B,C] into the table gradient for ALL B rows of `inputs`.
    The coordinate-only half of that scatter (histogram + scans of the binned backward) is issued here, in the forward, on a second
    stream: it overlaps the rest of the forward pass and the field backward instead of sitting on the critical path."""

    @staticmethod
    def forward(ctx, enc, inputs, embeddings, offsets_host, per_level_scale, base_resolution, gridtype, align_corners, interpolation, overlap,
                grad_in_place=False, plan=None):
        ctx.param = embeddings if grad_in_place else None
        L, B, C = enc.shape
        D = inputs.shape[1]
        S, H = float(np.log2(per_level_scale)), int(base_resolution)
        dt = dtype_id(enc)
        ctx.save_for_backward(inputs)
        ctx.cfg = (offsets_host, B, D, C, L, S, H, gridtype, interpolation, align_corners, tuple(embeddings.shape))
        if plan is not None and (plan.inputs_ptr != inputs.data_ptr() or plan.rows != B):
            plan = None                                                          # prepared for other coordinates: not ours
        if plan is None and overlap:
            plan = _prepare_plan(inputs, offsets_host, B, D, C, L, S, H, gridtype, align_corners, interpolation, dt, enc.device)
        ctx.plan = plan
        return enc.detach()

    @staticmethod
    def backward(ctx, grad):
        inputs, = ctx.saved_tensors
        offsets_host, B, D, C, L, S, H, gridtype, interpolation, align_corners, eshape = ctx.cfg
        grad = grad.contiguous()
        if _PRE_SCATTER_HOOK is not None:
            _PRE_SCATTER_HOOK()
        # grad_in_place (the trainers' persistent flat .grad buffer): the scatter is a read-modify-write of its destination anyway, so it
        # accumulates straight into embeddings.grad — no 49 MB zero-fill and no AccumulateGrad add pass (~45 us per step at T = 2^19)
        tgt = ctx.param.grad if ctx.param is not None else None
        in_place = (tgt is not None and tgt.dtype == torch.float32 and tgt.is_contiguous() and tuple(tgt.shape) == tuple(eshape)
                    and tgt.device == grad.device)
        grad_embeddings = tgt if in_place else torch.zeros(eshape, device=grad.device, dtype=torch.float32)
        if in_place:
            grad_chain_wait(grad.device)                                        # read-modify-write of the shared .grad: one scatter at a time
        if ctx.plan is not None:
            ws, ev = ctx.plan.ws, ctx.plan.event
            torch.cuda.current_stream().wait_event(ev)
            try:
                check(lib.cnerf_grid_encode_backward_prepared(ptr(grad), ptr(inputs), offsets_host.ctypes.data, ptr(grad_embeddings), B, D, C, L, L, S, H,
                                                              gridtype, int(align_corners), interpolation, dtype_id(grad), ptr(ws), ws.numel(), stream()),
                      "grid_encode_backward_prepared")
            finally:
                ctx.plan = None                                                 # releases the side workspace for the next plan
        else:
            ws, ws_bytes = _bwd_workspace(offsets_host, B, D, C, L, L, S, H, dtype_id(grad), grad.device)
            check(lib.cnerf_grid_encode_backward(ptr(grad), ptr(inputs), offsets_host.ctypes.data, ptr(grad_embeddings), B, D, C, L, L, S, H, None, None,
                                                 gridtype, int(align_corners), interpolation, dtype_id(grad), ptr(ws), ws_bytes, stream()), "grid_encode_backward")
        if in_place:
            grad_chain_record(grad.device)
        return None, None, (None if in_place else grad_embeddings), None, None, None, None, None, None, None, None, None


def grid_encode(inputs, embeddings, offsets, per_level_scale, base_resolution, calc_grad_inputs=False, gridtype=0,
                align_corners=False, interpolation=0, max_level=None):
    """Functional form with the reference's signature (grid.py:27,99): returns [B, L*C]."""
    table = embeddings
    if torch.is_autocast_enabled() and embeddings.shape[1] % 2 == 0:          # grid.py:45-46
        table = embeddings.detach().to(torch.half)
    out = _grid_encode.apply(inputs, embeddings, table.detach(), _offsets_host(offsets), per_level_scale, base_resolution,
                             calc_grad_inputs, gridtype, align_corners, interpolation, max_level)
    L, B, C = out.shape
    return out.permute(1, 0, 2).reshape(B, L * C)


class GridEncoder(nn.Module):
    """grid.py:102-192."""

    def __init__(self, input_dim=3, num_levels=16, level_dim=2, per_level_scale=2, base_resolution=16, log2_hashmap_size=19,
                 desired_resolution=None, gridtype='hash', align_corners=False, interpolation='linear'):
        super().__init__()
        if desired_resolution is not None:                                       # grid.py:107-108
            per_level_scale = np.exp2(np.log2(desired_resolution / base_resolution) / (num_levels - 1))
        if level_dim not in (1, 2, 4, 8):
            raise ValueError("GridEncoding: C must be 1, 2, 4, or 8.")            # gridencoder.cu:380
        if input_dim not in (2, 3, 4, 5):
            raise ValueError("GridEncoding: D must be 2, 3, 4, or 5.")            # gridencoder.cu:397
        self.input_dim = input_dim
        self.num_levels = num_levels
        self.level_dim = level_dim
        self.per_level_scale = per_level_scale
        self.log2_hashmap_size = log2_hashmap_size
        self.base_resolution = base_resolution
        self.output_dim = num_levels * level_dim
        self.gridtype = gridtype
        self.gridtype_id = _gridtype_to_id[gridtype]
        self.interpolation = interpolation
        self.interp_id = _interp_to_id[interpolation]
        self.align_corners = align_corners

        offsets, offset = [], 0                                                  # grid.py:124-135
        self.max_params = 2 ** log2_hashmap_size
        for i in range(num_levels):
            resolution = int(np.ceil(base_resolution * per_level_scale ** i))
            params_in_level = min(self.max_params, (resolution if align_corners else resolution + 1) ** input_dim)
            params_in_level = int(np.ceil(params_in_level / 8) * 8)
            offsets.append(offset)
            offset += params_in_level
        offsets.append(offset)
        self._offsets_host = np.array(offsets, dtype=np.int32)
        self.register_buffer('offsets', torch.from_numpy(self._offsets_host.copy()))
        self.n_params = offsets[-1] * level_dim
        self.embeddings = nn.Parameter(torch.empty(offset, level_dim))
        self._half_table = None
        self._half_version = None
        self.reset_parameters()

    def reset_parameters(self):
        std = 1e-4                                                                # grid.py:144-146
        self.embeddings.data.uniform_(-std, std)
        self.invalidate_half_table()                                              # a `.data` write does not move the version counter

    def __repr__(self):
        return (f"GridEncoder: input_dim={self.input_dim} num_levels={self.num_levels} level_dim={self.level_dim} "
                f"resolution={self.base_resolution} -> {int(round(self.base_resolution * self.per_level_scale ** (self.num_levels - 1)))} "
                f"per_level_scale={self.per_level_scale:.4f} params={tuple(self.embeddings.shape)} gridtype={self.gridtype} "
                f"align_corners={self.align_corners} interpolation={self.interpolation}")

    # ---- fp16 shadow table ------------------------------------------------------------------------------------
    def half_table(self):
        """fp16 copy of the table, refreshed when the parameter's version counter moved (optimizer step / load)."""
        emb = self.embeddings
        key = (emb._version, emb.data_ptr(), getattr(emb, '_cnerf_epoch', 0))
        if self._half_table is None or self._half_version != key or self._half_table.device != emb.device:
            if self._half_table is None or self._half_table.shape != emb.shape or self._half_table.device != emb.device:
                self._half_table = torch.empty(emb.shape, dtype=torch.half, device=emb.device)
            check(lib.cnerf_cast_f32_to_f16(ptr(emb), ptr(self._half_table), emb.numel(), stream()), "cast_f32_to_f16")
            self._half_version = key
        return self._half_table

    def invalidate_half_table(self):
        """Force the next half-precision gather to re-cast the table.  The shadow is keyed on the parameter's autograd version counter,
        which writes through `.data` do NOT move (reset_parameters, torch_ema's copy_to / restore, `param.data.copy_` in a drop-in
        trainer): whoever writes the table that way calls this afterwards (load_checkpoint and reset_parameters do)."""
        self._half_version = None

    def set_half_table(self, table, version_key=None):
        """Let a fused optimiser hand over the fp16 shadow it wrote while updating the parameter."""
        self._half_table = table
        emb = self.embeddings
        self._half_version = version_key if version_key is not None else (emb._version, emb.data_ptr(), getattr(emb, '_cnerf_epoch', 0))

    def encode(self, inputs, bound=1, max_level=None, half=None):
        """inputs [..., D] in [-bound, bound] -> kernel-layout features [L, B, C] (B = prod of leading dims)."""
        inputs = (inputs + bound) / (2 * bound)                                   # grid.py:156
        inputs = inputs.reshape(-1, self.input_dim)
        if half is None:
            half = torch.is_autocast_enabled() and self.level_dim % 2 == 0      # grid.py:45
        table = self.half_table() if half else self.embeddings.detach()
        return _grid_encode.apply(inputs, self.embeddings, table, self._offsets_host, self.per_level_scale, self.base_resolution,
                                  inputs.requires_grad, self.gridtype_id, self.align_corners, self.interp_id, max_level)

    @torch.no_grad()
    def encode_into(self, inputs_unit, out, row0, half=None, traversal=LEVEL_MAJOR):
        """Gather the features of `inputs_unit` [B, D] (already mapped to [0,1]) into rows row0 .. row0+B of the kernel-layout buffer
        `out` [L, P, C] (P >= row0 + B).  No autograd: pair with attach_backward once the buffer is complete.
        traversal: LEVEL_MAJOR (default), SAMPLE_MAJOR, or a TraversalTuner (the call site's; it measures both forms now and then)."""
        if half is None:
            half = torch.is_autocast_enabled() and self.level_dim % 2 == 0
        table = self.half_table() if half else self.embeddings.detach()
        require_cuda(inputs_unit, out, table)
        inputs_unit = inputs_unit.contiguous().float()
        B, D = inputs_unit.shape
        L, P, C = out.shape
        assert out.is_contiguous() and out.dtype == table.dtype and C == table.shape[1] and row0 + B <= P and L == self._offsets_host.shape[0] - 1
        prof = _PROFILE
        dst = out.data_ptr() + row0 * C * out.element_size()
        tuner = traversal if isinstance(traversal, TraversalTuner) else None
        todo = tuner.plan() if tuner is not None else [int(traversal)]
        for k, trav in enumerate(todo):
            timed = prof is not None or len(todo) > 1
            if timed:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            check(lib.cnerf_grid_encode_forward_ordered(ptr(inputs_unit), ptr(table), self._offsets_host.ctypes.data, dst, B, D, C, L, L,
                                                        float(np.log2(self.per_level_scale)), int(self.base_resolution), None, self.gridtype_id,
                                                        int(self.align_corners), self.interp_id, dtype_id(table), P, trav, stream()), "grid_encode_forward_ordered")
            if timed:
                e1.record()
                if len(todo) > 1:
                    tuner.record(trav, e0, e1)
                if prof is not None and k == len(todo) - 1:
                    prof.append((e0, e1, B, L, table.element_size()))

    def prepare_backward(self, inputs_unit, half):
        """Issue the coordinate-only half of the backward scatter for inputs_unit [P, D] (float32, contiguous, [0, 1] grid coordinates) NOW, on
        the side stream — as soon as the coordinates exist, e.g. before the last encode_into, whose gather it then overlaps — and return the
        plan for attach_backward(..., plan=).  None when nothing was issued (attach_backward then does it itself)."""
        assert inputs_unit.is_contiguous() and inputs_unit.dtype == torch.float32
        P, D = inputs_unit.shape
        dt = dtype_id(torch.empty(0, dtype=torch.float16 if half else torch.float32))
        return _prepare_plan(inputs_unit, self._offsets_host, P, D, self.level_dim, self.num_levels, float(np.log2(self.per_level_scale)),
                             int(self.base_resolution), self.gridtype_id, self.align_corners, self.interp_id, dt, inputs_unit.device)

    def hist_block_points(self, half):
        """granularity (rows) of prepare_backward_rows for this precision; 0 = the piecewise plan is not available (float32 records)"""
        import ctypes
        n = ctypes.c_uint32(0)
        check(lib.cnerf_grid_encode_backward_prepare_block(dtype_id(torch.empty(0, dtype=torch.float16 if half else torch.float32)), ctypes.addressof(n)),
              "grid_encode_backward_prepare_block")
        return int(n.value)

    def prepare_backward_rows(self, state, inputs_unit, half, row0, rows, finish):
        """prepare_backward in pieces: count rows [row0, row0 + rows) of inputs_unit [P, D] now (they must be written), `finish` on the last piece
        -> state to pass to the next call, the plan after the last one, or None (nothing issued; attach_backward then plans by itself)."""
        assert inputs_unit.is_contiguous() and inputs_unit.dtype == torch.float32
        P, D = inputs_unit.shape
        dt = dtype_id(torch.empty(0, dtype=torch.float16 if half else torch.float32))
        return _plan_rows(state, inputs_unit, self._offsets_host, P, D, self.level_dim, self.num_levels, float(np.log2(self.per_level_scale)),
                          int(self.base_resolution), self.gridtype_id, self.align_corners, self.interp_id, dt, inputs_unit.device, row0, rows, finish)

    def attach_backward(self, enc, inputs_unit, overlap=True, plan=None):
        """enc [L, P, C] filled by encode_into for the rows of inputs_unit [P, D] -> the same features, differentiable in the table.
        overlap: issue the coordinate-only half of the backward scatter now, on a second stream (see _grid_attach); plan: it was issued
        earlier by prepare_backward on these coordinates."""
        return _grid_attach.apply(enc, inputs_unit.contiguous().float(), self.embeddings, self._offsets_host, self.per_level_scale, self.base_resolution,
                                  self.gridtype_id, self.align_corners, self.interp_id, bool(overlap), bool(getattr(self, 'grad_in_place', False)), plan)

    def forward(self, inputs, bound=1, max_level=None, return_kernel_layout=False):
        """grid.py:151-168: [..., D] -> [..., L*C]."""
        prefix_shape = list(inputs.shape[:-1])
        out = self.encode(inputs, bound, max_level)
        if return_kernel_layout:
            return out
        L, B, C = out.shape
        return out.permute(1, 0, 2).reshape(prefix_shape + [self.output_dim])

    @torch.no_grad()
    def grad_total_variation(self, weight=1e-7, inputs=None, bound=1, B=1000000):
        """grid.py:171-192 (always float32)."""
        D, C = self.input_dim, self.embeddings.shape[1]
        L = self._offsets_host.shape[0] - 1
        S = float(np.log2(self.per_level_scale))
        H = self.base_resolution
        if inputs is None:
            inputs = torch.rand(B, self.input_dim, device=self.embeddings.device)
        else:
            inputs = ((inputs + bound) / (2 * bound)).view(-1, self.input_dim)
            B = inputs.shape[0]
        if self.embeddings.grad is None:
            raise ValueError('grad is None, should be called after loss.backward() and before optimizer.step()!')
        inputs = inputs.contiguous().float()
        require_cuda(inputs, self.embeddings)
        check(lib.cnerf_grad_total_variation(ptr(inputs), ptr(self.embeddings), ptr(self.embeddings.grad), self._offsets_host.ctypes.data,
                                             float(weight), B, D, C, L, S, H, self.gridtype_id, int(self.align_corners), stream()),
              "grad_total_variation")
