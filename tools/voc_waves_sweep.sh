for S in 256 512 1024; do for w in 8 4 3 2; do
 echo -n "S=$S waves=$w: "
 VP_VOC_WAVES=$w python bench.py --no-cpu --single-mode --mode voc --streams $S --steps 100 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,3), 'M frames/s', d['kernel_us'])"
done; done
