for cfg in "0.85 2" "0.7 2" "0.5 2" "0.0 2" "0.7 4" "0.5 4" "0.0 4" "0.85 1" "0.5 1"; do set -- $cfg
  PBR_SHADE_BIGFRAC=$1 PBR_SHADE_ROWS_SMALL=$2 python tools/shade_tile_ms.py 1928 2164 "bigfrac $1 rows_small $2" 2>&1 | grep shade
done
PBR_SHADE_BIGFRAC=0.85 python tools/shade_tile_ms.py 3840 2160 "4K bigfrac 0.85" 2>&1 | grep shade
PBR_SHADE_BIGFRAC=0.5 python tools/shade_tile_ms.py 3840 2160 "4K bigfrac 0.5" 2>&1 | grep shade
python tools/shade_tile_ms.py 7680 4320 "8K" 2>&1 | grep shade
