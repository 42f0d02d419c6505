for c in 0,0,2 0,0,1; do
  line="$c: base $(ONLY=$c python tools/conv_layers.py 2>/dev/null | grep kind | awk '{print $4}')"
  for b in 128 384 896 1024 1920; do
    v=$(NUNET_LIB_PATH=$GRAFT_REPO_ROOT/tools/_diag/libnunet_abl$b.so ONLY=$c timeout -k 10 100 python tools/conv_layers.py 2>/dev/null | grep kind | awk '{print $4}')
    line="$line | abl$b $v"
  done
  echo "$line"
done
