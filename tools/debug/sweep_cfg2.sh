for cfg in "0.85 2" "0.5 2" "0.0 4" "0.0 2" "0.0 1" "0.5 4"; do set -- $cfg
  PBR_SHADE_BIGFRAC=$1 PBR_SHADE_ROWS_SMALL=$2 python tools/cfg2_ms.py "bigfrac $1 rows_small $2" 2>&1 | grep Gpx
done
