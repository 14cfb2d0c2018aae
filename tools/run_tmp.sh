bash tools/measure_r06.sh a
