for mf in 0 1; do echo "== MFAST $mf shape 8 (512->512, 8x192x192: in 0.604 GB, out 0.604 GB)"; bash scripts/pmc_conv.sh mf${mf}_8 8 KOCR_W43_MFAST=$mf; done
for mf in 0 1; do echo "== MFAST $mf shape 9 (256->256, 8x384x384: in 1.208 GB, out 1.208 GB)"; bash scripts/pmc_conv.sh mf${mf}_9 9 KOCR_W43_MFAST=$mf; done
echo "== old kernel shape 8"; bash scripts/pmc_conv.sh old_8 8 KOCR_W43=0
