#!/usr/bin/env python3
"""Flat random-agent self-play for 3 / 4 players (row N4) as a stand-alone command, for rocprofv3:
    python3 tools/players_bench.py [--games 4096] [--chunk 256] [--launches 6] [--players 3 4] [--ext FLAGS]
Prints one JSON line per player count (env steps/s, average launch time from the event pairs around the launches)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--chunk", type=int, default=256)
    ap.add_argument("--launches", type=int, default=6)
    ap.add_argument("--players", type=int, nargs="+", default=[3, 4])
    ap.add_argument("--ext", type=int, default=0, help="extended-rule flags (azul_batch_create_rules), 0 = the reference's rules")
    ap.add_argument("--no-outputs", action="store_true")
    ap.add_argument("--mask-bits", action="store_true", help="also write the bit-packed mask stream")
    args = ap.parse_args()
    import torch
    from azul_deep_reinforcement_learning_amd import BatchedAzul
    for P in args.players:
        kw = {"ext_rules": args.ext} if args.ext else {}
        env = BatchedAzul(args.games, players=P, **kw)
        env.seed(0)
        env.init()
        env.new_round()
        bufs = env.alloc_trajectory(args.chunk, packed_mask=True, mask_pitch={5: 192, 7: 256, 9: 320}[env.displays], mask_bits=args.mask_bits)
        if args.no_outputs:
            run = lambda: env.selfplay(args.chunk)
        else:
            run = lambda: env.selfplay(args.chunk, bufs["mask"], bufs["action"], bufs["reward"], bufs["done"], maskbits=bufs.get("maskbits"), packed=bufs["packed"])
        run()
        torch.cuda.synchronize()
        stuck0 = int(env.counters()["stuck"].sum())
        t0 = time.perf_counter()
        env.timing_begin()
        for _ in range(args.launches):
            run()
        _, _, kms, kn = env.timing_end()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c = env.counters()
        moves = args.games * args.chunk * args.launches - (int(c["stuck"].sum()) - stuck0)
        print(json.dumps({"players": P, "ext": args.ext, "displays": env.displays, "num_actions": env.num_actions, "games": args.games, "moves_per_launch": args.chunk, "launches": args.launches,
                          "env_steps_per_s_wall": moves / dt, "avg_launch_ms": kms / max(kn, 1),
                          "env_steps_per_s_kernel": args.games * args.chunk / (kms / max(kn, 1) / 1e3),
                          "episodes": int(c["episodes"].sum()), "stuck": int(c["stuck"].sum())}), flush=True)
        del env, bufs
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
