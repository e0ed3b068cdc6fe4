#!/usr/bin/env python3
"""Runs the differential tester (tests/fuzz_plans.py) over a range of seeds on the GPU and prints every case whose
device answers differ from the oracle's.

    python tools/fuzz_device.py [--first S] [--count K] [--max-rows N]"""
import argparse
import os
import sys
import time

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--count", type=int, default=200)
    ap.add_argument("--max-rows", type=int, default=2_600_000)
    ap.add_argument("--only-after", default=None, help="only cases that end this way (finalize / blob / merge / ranks)")
    ap.add_argument("--repeat", type=int, default=1)
    ap.add_argument("--native-stacks", action="store_true", help="on a stuck case, also print rocgdb's view of every thread")
    ap.add_argument("--host-only", action="store_true", help="HOST buffers only, no threaded ranks, no torch")
    ap.add_argument("--seed-timeout", type=int, default=60, help="seconds before a case counts as stuck")
    ap.add_argument("--seconds", type=int, default=0, help="stop starting new cases after this long (0: run them all)")
    args = ap.parse_args()
    from fuzz_plans import Case, run_seed

    import faulthandler
    import signal

    def stuck(_sig, _frm):  # a case that hangs (threaded ranks waiting for each other) must not eat the GPU budget
        print("STUCK seed %d: %s" % (seed, Case(seed, args.max_rows, args.host_only).describe()), flush=True)
        faulthandler.dump_traceback(all_threads=True)
        if args.native_stacks:  # where the threads are inside libtgx / the HIP runtime (a debugger run by a helper
            import subprocess   # process this one has allowed to attach)
            import ctypes

            try:
                ctypes.CDLL(None).prctl(0x59616d61, -1, 0, 0, 0)  # PR_SET_PTRACER, PR_SET_PTRACER_ANY
                out = subprocess.run(["rocgdb", "-p", str(os.getpid()), "-batch", "-ex", "thread apply all bt 14"],
                                     capture_output=True, text=True, timeout=90)
                print(out.stdout[-12000:], flush=True)
            except Exception as e:  # noqa: BLE001
                print("no native stacks: %r" % (e,), flush=True)
        os._exit(3)

    import term_amd as T

    if not args.host_only:
        import torch  # noqa: F401  (the first import on a fresh box takes a minute or two: not a stuck case)

        torch.zeros(1, device="cuda")
    T.init()
    signal.signal(signal.SIGALRM, stuck)
    bad = 0
    t0 = time.time()
    seeds = list(range(args.first, args.first + args.count)) * args.repeat
    done = 0
    budget = args.seconds if args.seconds > 0 else 1e18
    for seed in seeds:
        if time.time() - t0 > budget:
            break
        # (a Case builds its tables: the filter is applied seed by seed, not over the whole range up front)
        if args.only_after is not None and Case(seed, args.max_rows, args.host_only).after != args.only_after:
            continue
        signal.alarm(args.seed_timeout)
        try:
            run_seed(seed, args.max_rows, args.host_only)
        except AssertionError as err:
            bad += 1
            print("FAIL", str(err).replace("\n", "\n     "), flush=True)
        done += 1
        if done % 500 == 0:
            print("... %d cases, %d failed, %.0f s" % (done, bad, time.time() - t0), flush=True)
    signal.alarm(0)
    print("%d cases of %d selected (seeds from %d), %d failed, %.0f s" % (done, len(seeds), args.first, bad, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
