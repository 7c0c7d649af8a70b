#!/bin/bash
. scripts/ab_lib.sh r06u
ab_layers "s3:" YOLOv4_608; ab_layers "s2:FASTMOT_STEM2=1" YOLOv4_608; ab_layers "s3b:" YOLOv4_608; ab_layers "s2b:FASTMOT_STEM2=1" YOLOv4_608
for f in s3 s2 s3b s2b; do head -6 $O/layers_YOLOv4_608_$f.txt | tail -3 | cut -c1-140; done
ab_bench 5 --steps 300 --warmup 10 -- "s3:" "s2:FASTMOT_STEM2=1"
