"""Margins of tests/test_full_size_gpu.py::test_benchmarked_epoch_matches_oracle: a whole epoch of the bench workloads on both
matrix pipes against the oracle, several seeds, with the band inside which rows are moved off the clip boundaries as a
parameter (0 = leave every row where it is: shows the bimodal deviation the band removes).
    python scratch/epoch_margin.py [band ...] > profiles/r4/epoch_margin.txt"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_full_size_gpu import run_epoch_against_oracle

bands = [float(b) for b in sys.argv[1:]] or [2e-5, 0.0]
shapes = [dict(name="doggo-4096env-2x256", D=58, A=12, H=256, N=4096, T=1000), dict(name="point-1024env-2x64", D=14, A=2, H=64, N=1024, T=2048)]
for band in bands:
    for shape in shapes:
        for seed, rs in ((23, 6), (24, 7), (25, 8)):
            t0 = time.time()
            errs, _, _, moved, passes = run_epoch_against_oracle(shape, seed, rs, band=band if band > 0 else 1e-30, log=lambda s: None)
            line = "  ".join(f"{pipe}: worst {max(e, key=e.get).replace('mlp_extractor.', '')} {max(e.values()):.2e} log_std {e['log_std']:.2e}"
                             for pipe, e in errs.items())
            print(f"band {band:g}  {shape['name']}  seed {seed}  moved {len(moved)} rows in {passes} pass(es)  {line}  ({time.time() - t0:.0f} s)", flush=True)
