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

    def finish(self):
        for s in (0, 1):
            self.wait_buffer_free(s)

    def gathered(self, slot, n_steps, games_per_rank):
        """[world][n_steps][games] views of the gathered records of `slot` (+ [..][3] mask words when shipped)."""
        res = {"packed": self.out[slot]["packed"].view(self.world, n_steps, games_per_rank)}
        if self.with_masks:
            res["maskbits"] = self.out[slot]["maskbits"].view(self.world, n_steps, games_per_rank, 3)
        return res
