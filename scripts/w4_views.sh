# usage (GPU box): bash scripts/w4_views.sh    F(2x2,3x3) vs F(4x4,3x3) (split over K where the grid is small) on the 3x3 layers by view count
for V in 1 2 4 8; do for a in winograd winograd4; do echo "== $a views $V"; python scripts/layer_time.py --views $V --$a --layers 4,6,8,10 2>/dev/null | grep "^L"; done; done
