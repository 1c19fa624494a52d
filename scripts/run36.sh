#!/bin/bash
REPS=5 STEPS=3 bash scripts/ab_bench.sh cornell 2>&1 | tail -13
