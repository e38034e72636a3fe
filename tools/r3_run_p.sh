#!/bin/bash
CCZ_TRACE_PHASES=2 python tools/offset_fit_probe.py 10 2>&1 | grep -v amdgpu.ids | tail -14
python tools/offset_fit_probe.py 0 2>&1 | grep offset | tail -4
