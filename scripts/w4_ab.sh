echo "== F(2x2,3x3)"; python scripts/layer_time.py --views 8 --winograd --layers 1,2,4,6,8,10,13,15,17,19,21 2>/dev/null | grep "^L\|^sum"
echo "== F(4x4,3x3)"; python scripts/layer_time.py --views 8 --winograd4 --layers 1,2,4,6,8,10,13,15,17,19,21 2>/dev/null | grep "^L\|^sum"
