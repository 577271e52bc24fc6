"""Data-parallel gradient exchange for the NeRF parameters — the layer the reference lacks (its DDP scaffolding is dead code:
nerf/utils_init_nerf.py:76-78, 709-726), designed for the xGMI topology of an 8 x MI355X node rather than for NCCL
habits: xGMI is point to point (7 links per GPU), so the exchange is an ALL-TO-ALL — every link carries its 1/world slice at the same
time — instead of a ring that is bound by one link.

Per step, for the big parameter (the hash-grid table: 12.2 M floats here, 47.9 M for the reference's bear field):

  1. pack      local float32 gradient * (1 / world) -> float16 payload (half the bytes; the same precision the reference's own fp16 training
               gives the table gradient: `__half2` atomics under the same GradScaler — gridencoder.cu:324-330), the float32 source is zeroed.
               The float16 payload exists ONLY under a dynamic loss scaler (scaled gradients, overflow found by the scaler's check of the
               reduced shard); float32 training and static loss scales send float32 — unscaled table gradients of 1e-7 and below would flush
               to zero in half precision;
  2. exchange  `all_to_all_single`: rank j receives slice j of every rank's payload (world x shard halves);
  3. reduce    the `world` slices are summed in FLOAT32 on arrival (no half-precision accumulation across ranks);
  4. update    Adam on the owned shard only (moments exist for the shard only: 1/world of the optimiser state per GPU), writing the
               float32 master shard and the float16 shadow shard in the same pass;
  5. publish   `all_gather_into_tensor` of the float16 SHADOW shards, in place in the shadow table — the forward pass reads the shadow,
               so 2 bytes per parameter come back instead of the 4 of an all-reduce.

The small parameters (the three MLPs: 22.5 k floats) are all-reduced in float32 and updated on every rank.  (`start_small` can be hooked between
the field backward and the grid scatter — `gridencoder.grid.set_pre_scatter_hook` — so that this all-reduce runs under the scatter kernels; that
needs `async_ops`, which is off: see __init__.)
With a dynamic loss scaler every rank checks its reduced shard (+ the MLP gradients) and the found-inf flags are OR-ed with one 4-byte
all-reduce(MAX), on the device: all ranks skip or step together, no host read.

Bytes per GPU and step (S = table floats): out 2 S (7/8 of it over the links), in 2 S (shadow) — against 8 S for a float32 all-reduce.

Works on any `torch.distributed` backend ("nccl" = RCCL on the GPUs; the world-2 gloo test in tests/test_dropin_and_dp.py runs the same
code on host tensors, where the Adam arithmetic is spelled out in torch — CUDA tensors always take the HIP kernel)."""
import torch
import torch.distributed as dist

from . import _coll

BIG_PARAM_MIN = 1 << 20
ALIGN = 64                      # shard boundaries in elements (float4 / half8 vector accesses of the Adam kernel, 128-byte lines)


def _adam_host(p, g, m, v, half_out, lr, betas, eps, step, inv_scale):
    """torch.optim.Adam's update on host tensors (gloo test only)"""
    g = g * inv_scale
    m.mul_(betas[0]).add_(g, alpha=1 - betas[0])
    v.mul_(betas[1]).addcmul_(g, g, value=1 - betas[1])
    bc1, bc2 = 1 - betas[0] ** step, 1 - betas[1] ** step
    p.addcdiv_(m, (v.sqrt() / bc2 ** 0.5).add_(eps), value=-lr / bc1)
    if half_out is not None:
        half_out.copy_(p)


class ShardedExchange:
    def __init__(self, params, flat, lr_of, world_size, rank, betas=(0.9, 0.99), eps=1e-15, scaler=None, half_shadow=False, group=None):
        """params: the trainable parameters in the order their .grad views sit in `flat` (trainer.flat_grad_buffer); lr_of(p) -> base lr."""
        self.world, self.rank, self.group = world_size, rank, group
        self.betas, self.eps, self.scaler = betas, eps, scaler
        self.flat = flat
        self.lr_of = lr_of
        off, self.big, self.small = 0, [], []
        for p in params:
            n = p.numel()
            (self.big if n >= BIG_PARAM_MIN else self.small).append((p, off, n))
            off += (n + 3) // 4 * 4
        if self.small:
            lo, hi = self.small[0][1], self.small[-1][1] + self.small[-1][2]
            assert all(o >= lo for _, o, _ in self.small) and not any(lo <= o < hi for _, o, _ in self.big), "small parameters must be contiguous in the flat buffer"
            self.small_seg = flat[lo:hi]
        else:
            self.small_seg = None
        self.payload_dtype = torch.float16 if scaler is not None else torch.float32
        self.state = []
        for p, o, n in self.big:
            s = ((n + world_size - 1) // world_size + ALIGN - 1) // ALIGN * ALIGN
            dev = p.device
            st = dict(p=p, off=o, n=n, shard=s,
                      send=torch.zeros(world_size * s, dtype=self.payload_dtype, device=dev),      # pad stays zero
                      recv=torch.empty(world_size, s, dtype=self.payload_dtype, device=dev),
                      g32=torch.empty(s, dtype=torch.float32, device=dev),
                      m=torch.zeros(s, dtype=torch.float32, device=dev), v=torch.zeros(s, dtype=torch.float32, device=dev),
                      # the float32 master of the OWNED shard (a padded private copy: the parameter's own storage stays the full table so that
                      # checkpoints / evaluation code see the usual tensor — consolidate() refreshes it)
                      master=torch.zeros(s, dtype=torch.float32, device=dev), step=0)
            lo, hi = rank * s, min((rank + 1) * s, n)
            if hi > lo:
                st['master'][:hi - lo].copy_(p.detach().reshape(-1)[lo:hi])
            if half_shadow:
                st['shadow'] = torch.zeros(world_size * s, dtype=torch.float16, device=dev)
                st['shadow'][:n].copy_(p.detach().reshape(-1))
            else:
                st['gather32'] = torch.zeros(world_size * s, dtype=torch.float32, device=dev)
            self.state.append(st)
        self._small_work = None
        # async_op=True collectives (handles waited on later) would let the MLP all-reduce run under the grid scatter, but with this torch / RCCL every
        # step that issues one runs ~3.5 % slower END TO END (edit step 16.5 -> 17.2 ms on one rank, uniformly over all kernels, UNet graph replay
        # included; the blocking form of the same collectives: 16.6 ms — scratch/edit_dp_world1.py).  Blocking collectives it is: "blocking" means
        # the compute stream waits for the collective, never the host.
        self.async_ops = False

    def describe(self):
        return (f"{'fp16' if self.payload_dtype == torch.float16 else 'fp32'} all-to-all of the table gradient (1/{self.world} pre-scaled, fp32 sum on arrival) + sharded Adam + all-gather of the "
                f"{'fp16 shadow' if 'shadow' in self.state[0] else 'fp32 master'} shards; MLP gradients fp32 all-reduce{' under the grid scatter' if self.async_ops else ''}")

    def shadow_table(self, p):
        """the full float16 shadow of big parameter p (what the forward gather reads), kept current by step()"""
        for st in self.state:
            if st['p'] is p and 'shadow' in st:
                return st['shadow'][:st['n']].view(p.shape)
        return None

    # ---- called from the backward pass, between the field backward and the grid scatter
    def start_small(self):
        if self.small_seg is not None and self._small_work is None:
            self._small_work = _coll.all_reduce(self.small_seg, op=dist.ReduceOp.SUM, group=self.group, async_op=self.async_ops)
            if self._small_work is None:
                self._small_work = True

    # ---- after the backward pass
    def exchange(self):
        """steps 1-3 for every big parameter; the small parameters' all-reduce is completed (started here if no hook did)"""
        inv_world = 1.0 / self.world
        works = []
        for st in self.state:
            src = self.flat[st['off']:st['off'] + st['n']]
            if src.is_cuda and self.payload_dtype == torch.float16:      # one pass: float32 -> pre-scaled float16 payload, source zeroed
                from ._lib import lib, check, ptr, stream
                check(lib.cnerf_dp_pack(ptr(src), ptr(st['send']), st['n'], inv_world, stream()), "dp_pack")
            else:
                torch.mul(src, inv_world, out=st['send'][:st['n']])      # pre-scaled so that the sum stays in range
                src.zero_()                                              # the scatter of the next step accumulates into it
            works.append(_coll.all_to_all_single(st['recv'].view(-1), st['send'], group=self.group, async_op=self.async_ops))
        self.start_small()
        for st, w in zip(self.state, works):
            if w is not None:
                w.wait()
            if st['recv'].is_cuda and self.payload_dtype == torch.float16:   # float32 accumulation on arrival (+ the scaler's found-inf test of the shard)
                from ._lib import lib, check, ptr, stream
                check(lib.cnerf_dp_reduce(ptr(st['recv']), self.world, st['shard'], ptr(st['g32']),
                                          ptr(self.scaler.state) if self.scaler is not None else None, stream()), "dp_reduce")
            else:
                torch.sum(st['recv'], dim=0, dtype=torch.float32, out=st['g32'])
        if self._small_work is not None and self._small_work is not True:
            self._small_work.wait()
        self._small_work = None

    def check(self):
        """found_inf over the owned shards and the reduced small gradients, OR-ed across ranks on the device"""
        if self.scaler is None:
            return
        for st in self.state:
            if not (st['g32'].is_cuda and self.payload_dtype == torch.float16):      # on the device cnerf_dp_reduce already tested the shard
                self.scaler.check(st['g32'])
        if self.small_seg is not None:
            self.scaler.check(self.small_seg)
        _coll.all_reduce(self.scaler.state[2:3], op=dist.ReduceOp.MAX, group=self.group)

    def step(self, lr_factor, loss_scale=1.0):
        """steps 4-5 for the big parameters.  The gradients in g32 are MEANS over ranks of loss_scale-scaled gradients (the 1/world factor went
        into the payload); the small parameters are left to the caller's optimiser (their reduced gradient is a SUM: un-scale by 1/(scale*world))."""
        for st in self.state:
            p, s, n = st['p'], st['shard'], st['n']
            lo = self.rank * s
            half_out = st['shadow'][lo:lo + s] if 'shadow' in st else None
            lr = self.lr_of(p) * lr_factor
            st['step'] += 1
            if p.is_cuda:
                from ._lib import lib, check, ptr, stream
                if self.scaler is not None:
                    check(lib.cnerf_adam_step_scaled(ptr(st['master']), ptr(st['g32']), ptr(st['m']), ptr(st['v']), ptr(half_out), s, float(lr),
                                                     float(self.betas[0]), float(self.betas[1]), float(self.eps), ptr(self.scaler.state), 1.0, 0, stream()),
                          "adam_step_scaled")
                else:
                    check(lib.cnerf_adam_step(ptr(st['master']), ptr(st['g32']), ptr(st['m']), ptr(st['v']), ptr(half_out), s, float(lr),
                                              float(self.betas[0]), float(self.betas[1]), float(self.eps), st['step'], 1.0 / loss_scale, 0, stream()), "adam_step")
            else:
                _adam_host(st['master'], st['g32'], st['m'], st['v'], half_out, lr, self.betas, self.eps, st['step'], 1.0 / loss_scale)
            if 'shadow' in st:
                _coll.all_gather_into_tensor(st['shadow'], st['shadow'][lo:lo + s], group=self.group)          # in place: 2 bytes per parameter
                p._cnerf_stale = True            # the float32 nn.Parameter now lags the owners' master shards until consolidate() (checkpoint.py refuses it)
            else:
                _coll.all_gather_into_tensor(st['gather32'], st['master'], group=self.group)
                p.data.reshape(-1).copy_(st['gather32'][:n])
            p._cnerf_epoch = getattr(p, '_cnerf_epoch', 0) + 1

    # ---- checkpoints: the optimiser state of a sharded run in the UNSHARDED layout (a checkpoint then loads into any world size)
    @torch.no_grad()
    def export_optimizer_state(self, optimizer):
        """gather the owners' Adam moments of the big parameters into `optimizer.state[p]` ({'step', 'exp_avg', 'exp_avg_sq'}: FusedAdam's /
        torch.optim.Adam's entry for p) and refresh the float32 parameters: `optimizer.state_dict()` afterwards is what an unsharded run
        would save (utils_init_nerf.py:793-794 stores the optimiser in 'full' checkpoints).  Collective: every rank calls it."""
        self.consolidate()
        for st in self.state:
            p, n, dev = st['p'], st['n'], st['p'].device
            full = torch.empty(self.world * st['shard'], dtype=torch.float32, device=dev)
            entry = optimizer.state[p]
            for key, src in (('exp_avg', st['m']), ('exp_avg_sq', st['v'])):
                _coll.all_gather_into_tensor(full, src, group=self.group)
                entry[key] = full[:n].clone().view(p.shape)
            entry['step'] = st['step']

    @torch.no_grad()
    def import_optimizer_state(self, optimizer, drop=True):
        """the inverse, after `optimizer.load_state_dict(...)` and the model's `load_state_dict`: this rank's slice of the loaded moments,
        the master shard and the shadow are taken from the full tensors; `drop` frees the full moments again."""
        for st in self.state:
            p, n, s = st['p'], st['n'], st['shard']
            lo, hi = self.rank * s, min((self.rank + 1) * s, n)
            entry = optimizer.state.get(p, {})
            for key, dst in (('exp_avg', st['m']), ('exp_avg_sq', st['v'])):
                dst.zero_()
                if key in entry and hi > lo:
                    dst[:hi - lo].copy_(entry[key].reshape(-1)[lo:hi])
            st['step'] = int(entry.get('step', 0))
            st['master'].zero_()
            if hi > lo:
                st['master'][:hi - lo].copy_(p.detach().reshape(-1)[lo:hi])
            if 'shadow' in st:
                st['shadow'][:n].copy_(p.detach().reshape(-1))
            if drop and p in optimizer.state:
                del optimizer.state[p]
            p._cnerf_epoch = getattr(p, '_cnerf_epoch', 0) + 1

    @torch.no_grad()
    def consolidate(self):
        """refresh the full float32 parameters from the owners' master shards (before a checkpoint / an evaluation that reads float32)"""
        for st in self.state:
            if 'shadow' not in st:
                continue                                                 # float32 mode keeps the parameter current in step()
            full = torch.empty(self.world * st['shard'], dtype=torch.float32, device=st['p'].device)
            _coll.all_gather_into_tensor(full, st['master'], group=self.group)
            st['p'].data.reshape(-1).copy_(full[:st['n']])
            st['p']._cnerf_stale = False
