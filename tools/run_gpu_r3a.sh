# round 3: per-workgroup timelines of the frame's GEMM classes (tools/ubench/gemm_timeline.hip), warm and cold
mkdir -p gpurun_out
tools/ubench/bin/gemm_timeline 0 > gpurun_out/r3_gemm_timeline_warm.txt 2>&1
tools/ubench/bin/gemm_timeline 1 > gpurun_out/r3_gemm_timeline_cold.txt 2>&1
tail -40 gpurun_out/r3_gemm_timeline_cold.txt
