import re, sys
S = sys.argv[1]; fn = int(sys.argv[2]) if len(sys.argv) > 2 else 3
lines = open(S).read().split('\n')
ends = [i for i, l in enumerate(lines) if l.startswith('.Lfunc_end')]
start = ends[fn - 1] if fn > 0 else 0; end = ends[fn]
lab = {}
for i in range(start, end):
    m = re.match(r'^(\.LBB\d+_\d+):', lines[i])
    if m: lab[m.group(1)] = i
loops = []
for i in range(start, end):
    m = re.search(r'\s(s_cbranch\w+|s_branch)\s+(\.LBB\d+_\d+)', lines[i])
    if m and m.group(2) in lab and lab[m.group(2)] < i:
        loops.append((lab[m.group(2)], i, m.group(2)))
def stats(a, b):
    ins = [l.strip() for l in lines[a:b + 1] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    sc = sum(1 for l in ins if l.startswith('scratch_'))
    v64 = sum(1 for l in ins if re.match(r'v_\w+_f64', l))
    sl = sum(1 for l in ins if l.startswith('s_load'))
    ds = sum(1 for l in ins if l.startswith('ds_'))
    gl = sum(1 for l in ins if l.startswith(('global_', 'flat_')))
    wl = sum(1 for l in ins if 'v_writelane' in l or 'v_readlane' in l)
    call = sum(1 for l in ins if l.startswith('s_swappc'))
    return len(ins), sc, v64, sl, ds, gl, wl, call
print("function lines %d-%d" % (start, end), "total:", stats(start, end))
print("loops (start,end,label): n_ins scratch f64 s_load ds global lane-spill calls")
for a, b, l in sorted(loops):
    print(a, b, l, stats(a, b))
