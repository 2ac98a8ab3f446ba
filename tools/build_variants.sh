#!/bin/bash
# Builds the prepared kernel variants next to the product library (git-ignored *.so, they travel with gpurun):
#   libmi_queue.so  -DMI_K1_QUEUE_KERNEL=1   tile search as a work queue (run with MI_K1_QUEUE=1)
#   libmi_pipe.so   -DMI_K4_PIPE=1           entropy coder as a walker wave + a range-coder wave per tile
#   libmi_pipe3.so  -DMI_K4_PIPE=2           entropy coder in three stages: walker wave | four CDF-adapter waves (disjoint rows) | range-coder wave
#   libmi_pipek.so  -DMI_K4_PIPE=3           the same three stages as three kernels (record stream in HBM, no wave waits for another)
#   libmi_combo.so  queue + LDS diet (5 workgroups / CU) + three-kernel entropy stage together (run with MI_K1_QUEUE=1)
#   libmi_rect.so   -DMI_RECT_PART=1         rectangular partitions of 8x8 nodes (check against oracle/_build/liboracle_rect.so)
#   libmi_diet.so   -DMI_K1_LDS_DIET=1 -DMI_K1_WG_PER_CU=5   tile search in 32 480 B of LDS and 96 VGPRs: five workgroups per CU, or four + entropy / filter kernels beside them
#   libmi_diet4.so  -DMI_K1_LDS_DIET=1       the LDS diet alone (128 VGPRs, four workgroups per CU, 32 KB of LDS per CU left for other kernels)
set -e
cd "$(dirname "$0")/.."
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-variable"
hipcc $F -DMI_K1_QUEUE_KERNEL=1 -o cavif_rs_amd/libmi_queue.so cavif_rs_amd/csrc/mi_avif.hip -lz &
hipcc $F -DMI_K4_PIPE=1 -o cavif_rs_amd/libmi_pipe.so cavif_rs_amd/csrc/mi_avif.hip -lz &
hipcc $F -DMI_K4_PIPE=2 -o cavif_rs_amd/libmi_pipe3.so cavif_rs_amd/csrc/mi_avif.hip -lz &
hipcc $F -DMI_K4_PIPE=3 -o cavif_rs_amd/libmi_pipek.so cavif_rs_amd/csrc/mi_avif.hip -lz &
hipcc $F -DMI_K1_QUEUE_KERNEL=1 -DMI_K1_LDS_DIET=1 -DMI_K1_WG_PER_CU=5 -DMI_K4_PIPE=3 -o cavif_rs_amd/libmi_combo.so cavif_rs_amd/csrc/mi_avif.hip -lz &
hipcc $F -DMI_RECT_PART=1 -o cavif_rs_amd/libmi_rect.so cavif_rs_amd/csrc/mi_avif.hip -lz &
hipcc $F -DMI_K1_LDS_DIET=1 -DMI_K1_WG_PER_CU=5 -o cavif_rs_amd/libmi_diet.so cavif_rs_amd/csrc/mi_avif.hip -lz &
hipcc $F -DMI_K1_LDS_DIET=1 -o cavif_rs_amd/libmi_diet4.so cavif_rs_amd/csrc/mi_avif.hip -lz &
wait
make -s -C oracle rect
ls -la cavif_rs_amd/libmi_*.so oracle/_build/
