for c in 4 8 16 32 64; do for d in 2 3 4; do timeout -k 10 60 build/batch_bench 1024,16384,131072 4096 16384 $c $d || exit 1; done; done
