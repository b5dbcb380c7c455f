# the reference's Re = 40 Newton example against the digitised figure: this build's protocol and one-ingredient variants
cd $GRAFT_REPO_ROOT
for v in "" "jactol=0.05" "jaccfl=0.4" "replay-literal" "replay"; do
  echo "=================== variant: ${v:-default}"; timeout -k 10 300 python3 scripts/cylinder_newton_re40.py $v 2>&1 | tail -62
done > gpurun_out/r04_re40_variants.txt 2>&1
tail -5 gpurun_out/r04_re40_variants.txt
