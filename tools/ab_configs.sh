#!/bin/bash
# usage (GPU box): tools/ab_configs.sh <variant .so> [repeats]   -> gpurun_out/ab_configs.txt
# The main build against a variant library on BASELINE.json's configurations, interleaved on one box.
lib=$1; reps=${2:-2}
cd $GRAFT_REPO_ROOT
H="RT_AMD_LIB=$PWD/$lib"
: > gpurun_out/ab_configs.txt
one() { tag=$1; shift; tools/ab.sh $tag "$H" "RT_MAIN=1" $reps -- "$@" > /dev/null; cut -c1-12,75-300 gpurun_out/ab_$tag.txt >> gpurun_out/ab_configs.txt; }
one mesh --steps 16 --warmup 3
one cornell --scene cornell_box --width 1920 --height 1080 --steps 16 --warmup 2
one monkey --scene mesh_monkey --width 3840 --height 2160 --steps 16 --warmup 2
one torus --scene torus --width 3840 --height 2160 --steps 16 --warmup 2
one csg --scene csg --width 3840 --height 2160 --steps 16 --warmup 2
one three --scene three-spheres --width 3840 --height 2160 --steps 8 --warmup 2
one room --scene room --width 3840 --height 2160 --steps 4 --warmup 1
one complex --scene complex_v2 --width 3840 --height 2160 --steps 4 --warmup 1
one fog --scene fog_light --width 7680 --height 4320 --samples 16 --steps 3 --warmup 1
cat gpurun_out/ab_configs.txt
