"""Multi-GPU: games shard by GLOBAL id (rank r owns games [G r, G (r+1)), seeds follow the global id, so a
game's trajectory does not depend on the GPU count) and the path's one exchange step is the all-gather of the
per-rank trajectory buffers.  One process per GPU, torch.distributed ("nccl" = RCCL over xGMI;
"gloo" in the CPU tests).

The all-gather is issued asynchronously right after the self-play launch that filled a buffer: RCCL runs it on
its own stream, concurrently with the NEXT launch, which writes the other buffer of a double-buffered pair.
xGMI is point-to-point, so the record is shipped compact (bit-packed mask: 33 B per env move instead of 189 B).
"""
import torch
import torch.distributed as dist


def shard_seed_base(seed_base, games_per_rank, rank):
    """Seed of a rank's first game: seeds are a function of the global game id."""
    return int(seed_base) + int(games_per_rank) * int(rank)


class TrajectoryGather:
    def __init__(self, world_size, device, group=None):
        self.world = int(world_size)
        self.device = device
        self.group = group
        self.work = [None, None]
        self.out = [None, None]
        self.gathered_bytes = 0

    def wait_buffer_free(self, slot):
        w = self.work[slot]
        if w is not None:
            w.wait()                 # the consumer stream waits for that gather; the host does not block on GPU work
            self.work[slot] = None

    def launch(self, slot, buf, n_steps):
        flat = buf["flat"]
        if self.out[slot] is None or self.out[slot].numel() != flat.numel() * self.world:
            self.out[slot] = torch.empty(flat.numel() * self.world, dtype=flat.dtype, device=flat.device)
        self.work[slot] = dist.all_gather_into_tensor(self.out[slot], flat, group=self.group, async_op=True)
        self.gathered_bytes += flat.numel() * self.world
        return self.out[slot]

    def finish(self):
        for s in (0, 1):
            self.wait_buffer_free(s)

    def gathered(self, slot, n_steps, games_per_rank):
        """Views [world][n_steps][games] into the gathered compact records of `slot`."""
        per = self.out[slot].numel() // self.world
        n, g = n_steps, games_per_rank
        sizes = [n * g * 24, n * g * 4, n * g * 4, n * g]
        res = {"maskbits": [], "action": [], "reward": [], "done": []}
        for r in range(self.world):
            chunk = self.out[slot][r * per:(r + 1) * per]
            o = 0
            res["maskbits"].append(chunk[o:o + sizes[0]].view(torch.int64).view(n, g, 3)); o += sizes[0]
            res["action"].append(chunk[o:o + sizes[1]].view(torch.int32).view(n, g)); o += sizes[1]
            res["reward"].append(chunk[o:o + sizes[2]].view(torch.int32).view(n, g)); o += sizes[2]
            res["done"].append(chunk[o:o + sizes[3]].view(n, g))
        return {k: torch.stack(v) for k, v in res.items()}
