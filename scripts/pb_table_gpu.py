"""Prints the device chain's numbers beside the reference's conv-poisson-boltzmann-harmonic-2d-rev390.txt rows
(tests/test_gpu_reference_tables.py asserts them).  usage on the GPU box: python scripts/pb_table_gpu.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import isph_amd  # noqa: F401
from isph_amd import hip
import pb_harmonic
from test_gpu_reference_tables import device_chain

ctx = hip.Context(0)
ref = pb_harmonic.known_answers()
for N in (16, 32, 64, 128, 256, 512, 1024):
    r = device_chain(ctx, N)
    print("N = %4d  newton %d, FGMRES+SA-AMG iterations %s, |F| %.1e" % (N, r["newton"], r["gmres"], r["residual"]))
    for name, key in (("total volume", "volume"), ("err.psi.norm2", "err_psi"), ("err.psi.grad.norm2", "err_grad")):
        print("    %-20s device %.15e   reference %.15e   rel. diff %.1e" % (name, r[key], ref[N][key], abs(r[key] - ref[N][key]) / ref[N][key]))

from test_gpu_reference_tables import device_channel
import pb_channel
for boundary in ("MorrisHolmes", "ConstExtension"):
    refc = pb_channel.known_answers(boundary)
    print("conv-channel-edl-potential-2d-morrisholmes-rev722.txt, section", boundary)
    for N in (32, 64, 128, 256, 512, 1024):
        r = device_channel(ctx, N, boundary)
        print("N = %4d  %d fluid particles, FGMRES+SA-AMG iterations %d" % (N, r["particles"], r["iters"]))
        for name, key in (("total volume", "volume"), ("err.psi.norm2", "err_psi")):
            print("    %-20s device %.15e   reference %.15e   rel. diff %.1e" % (name, r[key], refc[N][key], abs(r[key] - refc[N][key]) / refc[N][key]))
