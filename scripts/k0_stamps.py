"""Measurement: per-phase wave cycles of k_sampler_shuffle_par (a -DRT_K0_STAMP build of librtx_hip.so; RTX_K0_REPORT=1 prints them)."""
import sys, time
sys.path.insert(0, ".")
from rustracer_amd import host
host.sampler_tables(1024, 4, 0, 64)
t = time.time(); host.sampler_tables(1024, 4, 0, 32768); print("32768 pixels x 8 tables:", round((time.time() - t) * 1e3, 1), "ms incl. copies")
