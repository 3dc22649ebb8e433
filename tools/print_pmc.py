"""One line per kernel from a tools/summarize_pmc.py JSON: VALU instructions per wave, stall shares, LDS conflicts, HBM bytes."""
import json
import sys

d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if k.startswith("_") or not isinstance(v, dict) or not v.get("SQ_WAVE_CYCLES", 0):
        continue
    wc = v["SQ_WAVE_CYCLES"]
    waves = max(v.get("SQ_WAVES", 1), 1)
    busy = max(v.get("SQ_BUSY_CYCLES", 1), 1)
    print(f"{k.replace('void ', '')[:40]:40s} waves {waves:8.0f} VALU/wave {v.get('SQ_INSTS_VALU', 0) / waves:7.0f} LDS/wave {v.get('SQ_INSTS_LDS', 0) / waves:6.0f} "
          f"VMEM/wave {v.get('SQ_INSTS_VMEM', 0) / waves:5.0f} wait_any {v.get('SQ_WAIT_ANY', 0) / wc:4.2f} wait_inst {v.get('SQ_WAIT_INST_ANY', 0) / wc:4.2f} "
          f"valu_busy {v.get('SQ_ACTIVE_INST_VALU', 0) / busy / 4:4.2f} lds_conf {v.get('SQ_LDS_BANK_CONFLICT', 0) / max(v.get('SQ_LDS_IDX_ACTIVE', 1), 1):4.2f} "
          f"gui_cycles {v.get('GRBM_GUI_ACTIVE', 0):9.0f} FETCH {v.get('FETCH_SIZE', 0) / 1024:7.1f}MB WRITE {v.get('WRITE_SIZE', 0) / 1024:7.1f}MB")
