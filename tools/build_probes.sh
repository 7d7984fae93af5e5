#!/bin/bash
# builds the HIP micro-benchmarks of tools/ into tools/bin/ (in the container: hipcc cross-compiles; the binaries travel with gpurun)
cd $(dirname $0)/..
mkdir -p tools/bin
I=-Ibayesiandatafusion.jl_amd/csrc
for p in fp64_pipe_probe cu_mask_probe gather_probe hip_call_cost; do hipcc --offload-arch=gfx950 -O3 -w -o tools/bin/$p tools/$p.hip; done
hipcc --offload-arch=gfx950 -O3 -w $I -o tools/bin/factor_probe tools/factor_probe.hip
hipcc --offload-arch=gfx950 -O3 -w $I -DPROBE_DP=64 -o tools/bin/factor_probe64 tools/factor_probe.hip
hipcc --offload-arch=gfx950 -O3 -w $I -DPROBE_BLOCKED -o tools/bin/factor_probe_b tools/factor_probe.hip
hipcc --offload-arch=gfx950 -O3 -w $I -DPROBE_BLOCKED -DPROBE_DP=64 -o tools/bin/factor_probe64_b tools/factor_probe.hip
hipcc --offload-arch=gfx950 -O3 -w -o tools/bin/valu_cost_probe tools/valu_cost_probe.hip
ls -la tools/bin
