"""Multi-GPU: games shard by GLOBAL id (rank r owns games [G r, G (r+1)), seeds follow the global id, so a
game's trajectory does not depend on the GPU count) and the path's one exchange step is the all-gather of the
per-rank trajectory buffers.  One process per GPU, torch.distributed ("nccl" = RCCL over xGMI;
"gloo" in the CPU tests).

What is shipped is the compact per-move record the kernel writes (`packed`: action | done << 8 | reward << 16,
4 bytes per move) and, optionally, the bit-packed legal mask (24 bytes per move).  Sizing: at ~2.1 G moves/s per GPU
the compact record is ~8.5 GB/s per GPU -- two orders of magnitude under an xGMI link (7 x ~153 GB/s per GPU, point to
point): 59 MB into each GPU per 512-move chunk with 8 ranks, ~0.2 ms against 1.0 ms of compute -- while masks would add
~50 GB/s per GPU, i.e. ~350 GB/s of all-gather traffic INTO each of 8 GPUs.  The gather is therefore kept small enough to
cost little even when it serialises with the launches, and the masks (a deterministic function of seed + actions) are
only shipped on request.

The all-gather is issued asynchronously right after the self-play launch that filled a buffer pair; the next launch
writes the other pair (double buffering), and a buffer is reused only after its gather has completed.
"""
import torch
import torch.distributed as dist


def shard_seed_base(seed_base, games_per_rank, rank):
    """Seed of a rank's first game: seeds are a function of the global game id."""
    return int(seed_base) + int(games_per_rank) * int(rank)


def unpack_moves(packed):
    """int32 compact records -> (action int32 with -1 for none, done uint8, reward int32)."""
    p = packed.to(torch.int32)
    action = p & 0xFF
    action = torch.where(action == 0xFF, torch.full_like(action, -1), action)
    done = ((p >> 8) & 0xFF).to(torch.uint8)
    reward = p >> 16                      # arithmetic shift: sign-extends the 16-bit reward
    return action, done, reward


# ---- the full C1 record of one agent step (what NNRunner.run_episode keeps per step, nn_runner.py:17-47, and what
# NNRunner.train concatenates over episodes before ONE update, nn_runner.py:59-78) in a wire format for the all-gather BASELINE configs[4]
# names.  184 bytes per (step, game):
#   0   obs[136]      u8   the 136 observation integers (game_runner.py:65-72; all in 0..255: tile counts, flags, scores <= 240)
#   136 maskbits[24]  u8   the 180 legal-move bits, bit a & 7 of byte a >> 3 (= bit a & 63 of little-endian word a >> 6)
#   160 action        u8   0..179, 0xff = none (stuck slot)
#   161 done, 162 player to move, 163 zero
#   164 reward i32 | 168 value f32 | 172 log_prob f32 | 176 entropy f32 | 180 discounted return f32
# The default multi-GPU paths do NOT ship this (the records feed the rank that produced them; only the 328 KB gradient is
# exchanged): it exists so that the collective north_star names can be run and timed.
C1_BYTES = 184


def pack_c1(tr, n_steps, out=None):
    """One window of a PolicyRollout trajectory (dict of [T(+1)][G][..] tensors) -> uint8 [T][G][184].  Tensors in HBM are packed by ONE
    launch of the library's azul_pack_c1 kernel on the current stream (the producer in front of the trajectory all-gather: no torch op
    touches the data path); the torch restatement below only serves host tensors -- the world-size-2 gloo tests on CPU, and the GPU test
    that holds the kernel to it byte for byte."""
    T = int(n_steps)
    obs, mask = tr["obs"][:T], tr["mask"][:T]
    G = obs.shape[1]
    if out is None:
        out = torch.empty(T, G, C1_BYTES, dtype=torch.uint8, device=obs.device)
    if obs.is_cuda:
        import ctypes as C
        from . import _lib as L
        p = lambda t: C.c_void_p(t.data_ptr())
        need = {"obs": obs, "mask": mask, "player": tr["player"][:T], "action": tr["action"][:T], "reward": tr["reward"][:T], "done": tr["done"][:T],
                "value": tr["value"][:T], "log_prob": tr["log_prob"][:T], "entropy": tr["entropy"][:T], "returns": tr["returns"][:T]}
        want = {"obs": torch.float32, "mask": torch.uint8, "player": torch.uint8, "action": torch.int32, "reward": torch.int32, "done": torch.uint8}
        for k, t in need.items():
            if not t.is_contiguous() or t.dtype != want.get(k, torch.float32):
                raise ValueError("pack_c1: `%s` must be a contiguous %s tensor (the rollout's own buffers are)" % (k, want.get(k, torch.float32)))
        if not out.is_contiguous() or out.dtype != torch.uint8 or tuple(out.shape) != (T, G, C1_BYTES):
            raise ValueError("pack_c1: `out` must be a contiguous uint8 [T][G][%d] tensor" % C1_BYTES)
        with torch.cuda.device(obs.device):
            L.check(L.lib.azul_pack_c1(p(obs), p(mask), p(need["player"]), p(need["action"]), p(need["reward"]), p(need["done"]), p(need["value"]),
                                       p(need["log_prob"]), p(need["entropy"]), p(need["returns"]), T, G, p(out),
                                       C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream)))
        return out
    out[..., :136] = obs.to(torch.uint8)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=obs.device)
    bits = torch.zeros(T, G, 192, dtype=torch.int32, device=obs.device)
    bits[..., :180] = (mask != 0).to(torch.int32)
    out[..., 136:160] = (bits.view(T, G, 24, 8) * w).sum(-1).to(torch.uint8)
    a = tr["action"][:T].to(torch.int32)
    out[..., 160] = torch.where(a < 0, torch.full_like(a, 0xFF), a).to(torch.uint8)
    out[..., 161] = tr["done"][:T].to(torch.uint8)
    out[..., 162] = tr["player"][:T].to(torch.uint8)
    out[..., 163] = 0
    words = out[..., 164:184].view(torch.int32)
    words[..., 0] = tr["reward"][:T].to(torch.int32)
    f = words[..., 1:5].view(torch.float32)
    f[..., 0] = tr["value"][:T].reshape(T, G)
    f[..., 1] = tr["log_prob"][:T]
    f[..., 2] = tr["entropy"][:T]
    f[..., 3] = tr["returns"][:T]
    return out


def unpack_c1(rec):
    """uint8 [...][184] -> dict of tensors (obs f32 [...][136], mask u8 [...][180], action i32 with -1 for none, done, player, reward, value,
    log_prob, entropy, returns)."""
    rec = rec.contiguous()
    lead = rec.shape[:-1]
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=rec.device)
    bits = ((rec[..., 136:160].to(torch.int32).unsqueeze(-1) & w) != 0).to(torch.uint8).reshape(*lead, 192)[..., :180]
    a = rec[..., 160].to(torch.int32)
    words = rec[..., 164:184].contiguous().view(torch.int32)
    f = words[..., 1:5].contiguous().view(torch.float32)
    return {"obs": rec[..., :136].to(torch.float32), "mask": bits, "action": torch.where(a == 0xFF, torch.full_like(a, -1), a),
            "done": rec[..., 161], "player": rec[..., 162], "reward": words[..., 0], "value": f[..., 0], "log_prob": f[..., 1],
            "entropy": f[..., 2], "returns": f[..., 3]}


class TrajectoryGather:
    def __init__(self, world_size, device, group=None, with_masks=False):
        self.world = int(world_size)
        self.device = device
        self.group = group
        self.with_masks = with_masks
        self.work = [[], []]
        self.out = [{}, {}]
        self.gathered_bytes = 0

    def wait_buffer_free(self, slot):
        for w in self.work[slot]:
            w.wait()                 # the consumer stream waits for that gather; the host does not block on GPU work
        self.work[slot] = []

    def _gather(self, slot, name, t):
        flat = t.reshape(-1)
        o = self.out[slot].get(name)
        if o is None or o.numel() != flat.numel() * self.world:
            o = torch.empty(flat.numel() * self.world, dtype=flat.dtype, device=flat.device)
            self.out[slot][name] = o
        self.work[slot].append(dist.all_gather_into_tensor(o, flat, group=self.group, async_op=True))
        self.gathered_bytes += o.numel() * o.element_size()

    def launch(self, slot, buf, n_steps):
        self._gather(slot, "packed", buf["packed"])
        if self.with_masks:
            self._gather(slot, "maskbits", buf["maskbits"])

    def launch_c1(self, slot, records):
        """Opt-in (bench.py --gather-c1): all-gather a window's full C1 records (pack_c1: [T][G][184] bytes per rank)."""
        self._gather(slot, "c1", records)

    def gathered_c1(self, slot, n_steps, games_per_rank):
        """[world][n_steps][games][C1_BYTES] view of the gathered C1 records of `slot`."""
        return self.out[slot]["c1"].view(self.world, n_steps, games_per_rank, C1_BYTES)

    def finish(self):
        for s in (0, 1):
            self.wait_buffer_free(s)

    def gathered(self, slot, n_steps, games_per_rank):
        """[world][n_steps][games] views of the gathered records of `slot` (+ [..][3] mask words when shipped)."""
        res = {"packed": self.out[slot]["packed"].view(self.world, n_steps, games_per_rank)}
        if self.with_masks:
            res["maskbits"] = self.out[slot]["maskbits"].view(self.world, n_steps, games_per_rank, 3)
        return res
