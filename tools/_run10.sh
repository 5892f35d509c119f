for tree in new old new old; do
  if [ $tree = old ]; then cd ab_old; fi
  echo "== $tree"; python tools/conv_bench.py f32 f16x3 2>/dev/null | cut -c1-110
  if [ $tree = old ]; then cd ..; fi
done
