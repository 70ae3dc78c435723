"""The wire format of the opt-in C1 all-gather (bench.py --gather-c1; parallel.pack_c1 / unpack_c1) on the KERNEL's trajectories: a
window recorded by azul_batch_policy_rollout on the GPU packs to 184 bytes per agent step and unpacks to exactly what was recorded --
observations (integers in 0..255 for every state the rules reach), the 180 mask bits, action / done / player / reward and the four floats
bit for bit -- and the pack on HBM tensors equals the pack of the same tensors on the CPU (what the gloo test of tests/test_distributed_gloo.py
ships).  World size 1 over "nccl": the gather itself runs through RCCL once.
Reference: NNRunner.run_episode's per-step record, nn_runner.py:17-47; what NNRunner.train concatenates, nn_runner.py:59-78."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("opponent", [None, "random"])
def test_pack_c1_round_trips_a_recorded_window(opponent):
    from azul_deep_reinforcement_learning_amd import BatchedActorCritic, PolicyRollout
    from azul_deep_reinforcement_learning_amd.parallel import C1_BYTES, pack_c1, unpack_c1
    torch.manual_seed(3)
    T, G = 32, 256
    ro = PolicyRollout(BatchedActorCritic(136, 180, 180), n_games=G, parts=1, window=T, persistent=True, opponent=opponent, seed_base=77)
    for _ in range(4):                                   # episodes end and restart inside these windows
        tr = ro.run_window()[0]
    ro.synchronize()
    rec = pack_c1(tr, T)
    assert rec.shape == (T, G, C1_BYTES) and rec.dtype == torch.uint8 and rec.is_cuda
    cpu = {k: v.cpu() for k, v in tr.items()}
    assert torch.equal(rec.cpu(), pack_c1(cpu, T))
    u = unpack_c1(rec)
    assert float(tr["obs"][:T].min()) >= 0 and float(tr["obs"][:T].max()) <= 255                 # what makes the u8 observation lossless
    assert torch.equal(u["obs"], tr["obs"][:T]) and torch.equal(u["mask"], tr["mask"][:T]) and torch.equal(u["player"], tr["player"][:T])
    for k in ("action", "reward", "done", "log_prob", "entropy", "returns"):
        assert torch.equal(u[k], tr[k][:T]), k
    assert torch.equal(u["value"], tr["value"][:T].reshape(T, G))
    assert int((tr["done"][:T] != 0).sum()) > 0


def test_c1_gather_runs_through_rccl_at_world_size_one():
    import torch.distributed as dist
    from azul_deep_reinforcement_learning_amd.parallel import C1_BYTES, TrajectoryGather
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        tg = TrajectoryGather(1, dev)
        rec = torch.randint(0, 256, (8, 16, C1_BYTES), dtype=torch.uint8, device=dev)
        tg.launch_c1(0, rec)
        tg.finish()
        torch.cuda.synchronize()
        assert torch.equal(tg.gathered_c1(0, 8, 16)[0], rec) and tg.gathered_bytes == rec.numel()
    finally:
        dist.destroy_process_group()
