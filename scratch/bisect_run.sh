for w in sphere norm ddn; do cp raytracer_amd/lib/librt_$w.so raytracer_amd/lib/librt_mi355x.so; echo "== $w"; timeout -k 10 120 python -u scratch/diag7.py | tail -4; done
