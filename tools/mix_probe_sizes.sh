for n in 2000 10000 50000 125000; do echo "N=$n"; timeout -k 10 200 python tools/mix_probe.py 20 $n 200; done
