#!/bin/bash
# Diagnostic library with -DGRACE_STAMPS (never the product): scratch/lib_stamps.so
cd "$(dirname "$0")/../../grace-devel_amd" && mkdir -p ../scratch && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -I../include -Icsrc -Wall -Wno-unused-function -fno-slp-vectorize -DGRACE_STAMPS $EXTRA -c csrc/trace.hip -o /tmp/trace_stamps.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../scratch/${OUT:-lib_stamps.so} /tmp/trace_stamps.o $(ls build/*.o | grep -v "build/trace.o")
