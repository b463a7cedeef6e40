import json,sys
d=json.load(open(sys.argv[1]))
for k,v in d.items():
    if isinstance(v,dict):
        m,c=v["matrix_wave_cycles"],v["copy_wave_cycles"]
        print(f"{k:36s} {v['ms_per_launch']:.3f} ms  K1 {m['kloop_1_with_deferred_epilogue_2']:6d} epi1 {m['epilogue_1_and_skip_init']:5d} K2 {m['kloop_2']:6d} waitA {m['wait_A']:5d} waitB {m['wait_B']:5d} | store {c['store_previous_board']:6d} writeX {c['write_X']:5d} cwaitA {c['wait_A']:6d} cwaitB {c['wait_B']:6d} {v['effective_GHz']:.2f} GHz")
