#!/bin/bash
T="c2 gen1@circle c2@circle c3@circle c3@edges c2-250k c2-dense"
B="b=optimized_f64 b=optimized_f32 b=voxel5_f64 b=tore_full_frame_f32 b=nimagenet_acc_all_f32"
echo "--- base"; timeout 600 python tools/sweep_table.py $T $B
echo "--- sb8"; EVREP_LIB_PATH=tools/variants/sb8.so timeout 600 python tools/sweep_table.py $T $B
