cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for W in speech_mel_b256x1x160000_nfft400_hop160_mel80 cfg4_stft_b128x1x110250_nfft1024_hop256_pad cfg4_istft_b128x1x434f_nfft1024_hop256 reftest_logmel_db_b256x2x22050_nfft512_hop128_mel40; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/qp_$W -- python $R/tools/pmc_run.py $W > /dev/null 2> $R/gpurun_out/qp_$W.log
  f=$(find $R/gpurun_out/qp_$W -name "*kernel_stats.csv" | head -1)
  echo "== $W"; head -8 $f | cut -d, -f1-6
done
