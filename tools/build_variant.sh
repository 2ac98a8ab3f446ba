#!/bin/bash
# tools/build_variant.sh NAME [extra hipcc flags]  -> cavif_rs_amd/libmi_v_NAME.so (git-ignored; travels with gpurun): an experiment build of the CURRENT sources,
# by default only the headline configuration's kernel instantiations (-DMI_FAST_BUILD: a third of the compile time; pass -UMI_FAST_BUILD for everything).
# A/B on the GPU: tools/ab.sh cavif_rs_amd/libmi_v_A.so cavif_rs_amd/libmi_v_B.so
NAME=$1; shift
cd "$(dirname "$0")/.."
exec hipcc --offload-arch=gfx950 -O2 -mllvm -sink-insts-to-avoid-spills -mllvm -inline-threshold=1000 -mllvm -disable-machine-licm -mllvm -phi-node-folding-threshold=1 -std=c++17 -ffp-contract=off -fPIC -shared -Wno-unused-variable -Wno-pass-failed -DMI_FAST_BUILD -DMI_K1_POLL_LOG2=20 "$@" -o cavif_rs_amd/libmi_v_$NAME.so ${SRC:-cavif_rs_amd/csrc}/mi_avif.hip -Iinclude -lz
