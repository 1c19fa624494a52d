#!/bin/bash
timeout 1200 python scripts/parity_report.py r03 2>&1 | grep -v amdgpu.ids
