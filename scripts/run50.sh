#!/bin/bash
timeout 1200 python scripts/soak_pbrt.py 2>&1 | grep -v amdgpu.ids | tail -5
timeout 600 python scripts/parity_report.py r03 2>&1 | grep -v amdgpu.ids | cut -c1-260
