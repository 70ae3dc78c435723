import cProfile, pstats, sys, random, io
sys.path.insert(0, ".")
sys.path.insert(0, "integration")
import torch
from azulnet.game_runner import GameRunner, RandomAgent
agent = RandomAgent()
def episode(seed):
    random.seed(seed)
    r = GameRunner(); r.reset()
    done = False
    while not done:
        mask = r.get_valid_moves()
        a = agent.get_a_output(None, torch.from_numpy(mask[None, :]))
        _, done = r.step(a)
episode(999)
pr = cProfile.Profile(); pr.enable()
for s in range(40): episode(s)
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(32); print(st.getvalue()[:7000])
