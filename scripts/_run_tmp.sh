P=$PWD/implementation_phd_lab_vision_amd
for v in s0 s1 s2 s3 sc1 sp4 sc1p4; do
echo "== stamp $v"; R50_TAIL3_VAR=1 R50_LIB=$P/libr50hip_$v.so timeout -k 10 200 python scripts/stamp_tail3p.py 2>&1 | grep -v amdgpu.ids
done
echo "== p4"; R50_LIB=$P/libr50hip_p4.so timeout -k 10 200 python scripts/time_tail3_variants.py 256 3 0,1 2>&1 | grep -v "amdgpu.ids\|equal"
