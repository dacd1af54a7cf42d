import sys, json
for l in sys.stdin:
    l = l.strip()
    if not l.startswith('{'):
        print(l); continue
    r = json.loads(l)
    print(f"{r['shape']:28s} M={r['M']:6d} N={r['N']:5d} K={r['K']:6d} auto=c{r['auto']}/s{r.get('auto_split','?')}  " +
          '  '.join(f"c{c}:{r.get(f'cfg{c}_us','-'):>7}us {r.get(f'cfg{c}_tf','-'):>6}TF" for c in (0, 1, 2, -1)))
