#!/bin/bash
T="gen1@circle gen1@edges c2@circle c3@circle c3@edges c2-250k c2-dense"
B="b=optimized_f64 b=optimized_f32 b=voxel5_f64 b=tore_full_frame_f32 b=nimagenet_acc_all_f32"
echo "--- base"; timeout 600 python tools/sweep_table.py $T $B
echo "--- defer0"; EVREP_LIB_PATH=tools/variants/defer0.so timeout 600 python tools/sweep_table.py $T $B
