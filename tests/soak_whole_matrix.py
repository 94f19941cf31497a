"""Not collected by pytest (no test_ prefix): the whole-matrix comparison of tests/test_gpu_whole_matrix.py with OTHER
seeds than the suite's 777 -- `python3 tests/soak_whole_matrix.py C3 T32 -- 1 2 3` on a GPU box.  It lives under tests/
because it drives the checker (oracle/nb_model)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    cut = args.index("--") if "--" in args else len(args)
    configs = args[:cut] or ["C3"]
    seeds = [int(a) for a in args[cut + 1:]] or [1, 2]
    from oracle import nb_model
    nb_model.install_hw_tables_from_device()
    assert nb_model.hw_mode()
    from test_gpu_whole_matrix import _whole_matrix_vs_model
    from prosstt_amd import workloads
    for name in configs:
        for seed in seeds:
            t0 = time.perf_counter()
            _whole_matrix_vs_model(name, workloads.CONFIGS[name]["N"], 5000, grouped=(name == "T32"), seed=seed)
            print("SOAK %s seed %d: every count equals the model's (%.0f s)" % (name, seed, time.perf_counter() - t0), flush=True)


if __name__ == "__main__":
    main()
