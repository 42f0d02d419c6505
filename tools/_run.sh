for c in 0 1; do
echo "== wide $c"
NUNET_CONV_WIDE=$c python tools/conv_layers.py 2>/dev/null | head -7
done
