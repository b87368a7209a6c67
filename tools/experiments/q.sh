echo "== stage64 (warm path for every unit > 64)"; EVREP_X_STAGE64=1 timeout 300 python tools/sweep_table.py gen1 c2-150k
echo "== default"; timeout 300 python tools/sweep_table.py gen1 c2-150k b=optimized_f64 b=event_stack_f32 b=time_surface_f64
