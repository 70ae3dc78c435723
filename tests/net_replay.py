"""TEST helper: replay what a rollout with a NETWORK opponent recorded for one game through the oracle's GameRunner(opponent=...)
(oracle/azul_oracle.c: oz_runner_step_with / oz_runner_reset_with, pinned to the reference by tests/golden/net_opponent.npz).
The opponent callback answers with the rollout's recorded opp_action trace and keeps what it was handed."""
import numpy as np

from oracle import oracle as oz


def replay_game(rec0, mt0, pos0, first, pool, action, opp_action, opp_replies, obs, mask, player, reward, done):
    """All arrays are this game's columns over the recorded steps: action / reward / done / opp_replies [S], opp_action [S][R],
    obs [S+1][136], mask [S+1][180], player [S+1].  Asserts every env-side record; returns (runner, handed) with handed =
    [(step, reply index, state int64[136], mask bool[180], player to move)] for every opponent call."""
    S, R = len(action), opp_action.shape[1]
    assert int(opp_replies.max(initial=0)) <= R, "a step had more replies than the trace holds: raise opponent_trace"
    cur = {"t": 0, "j": 0}
    handed = []

    def opponent(s, m):
        t, j = cur["t"], cur["j"]
        a = int(opp_action[t, j])
        assert 0 <= a < 180 and m[a], ("opponent answer not legal", t, j, a)
        handed.append((t, j, s.copy(), m.copy(), int(run.q.game.current_player)))
        cur["j"] += 1
        return a

    run = oz.NetRunner(opponent, first, pool, rec=rec0, mt=mt0, pos=pos0)
    for t in range(S):
        cur["t"], cur["j"] = t, 0
        m = run.get_valid_moves()
        assert np.array_equal(np.asarray(mask[t]).astype(bool), m), ("mask", t)
        assert np.array_equal(np.asarray(obs[t]).astype(np.int64), run.get_state(0)), ("obs", t)
        assert int(player[t]) == 1 == int(run.q.game.current_player) and int(m.sum()) >= 2, ("player", t)
        a = int(action[t])
        assert 0 <= a < 180 and m[a], ("agent action", t, a)
        rc, rew, dn = run.step(a)                                   # game_runner.py:43-55 with the recorded opponent
        assert rc == 0 and rew == int(reward[t]) and dn == bool(done[t]), ("reward / done", t, rew, int(reward[t]), dn, int(done[t]))
        if dn:
            assert run.reset() == 0                                 # nn_runner.py:20 -> game_runner.py:76-85: the opponent opens
        assert cur["j"] == int(opp_replies[t]), ("replies", t, cur["j"], int(opp_replies[t]))
    assert np.array_equal(np.asarray(mask[S]).astype(bool), run.get_valid_moves()) and np.array_equal(np.asarray(obs[S]).astype(np.int64), run.get_state(0))
    return run, handed
