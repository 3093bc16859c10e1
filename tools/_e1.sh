REPS=4 STEPS=100 bash tools/ab_graph.sh "fill1:X=1" "fill0:PASSION_MARCH_FILL=0" > gpurun_out/e1.txt 2>&1
