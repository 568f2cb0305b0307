import sys, textwrap, re
W=150
def split_cells(line):
    line=line.strip()
    assert line.startswith('|')
    cells=[]; cur=''; tick=False
    for ch in line[1:]:
        if ch=='`': tick=not tick
        if ch=='|' and not tick:
            cells.append(cur.strip()); cur=''
        else: cur+=ch
    if cur.strip(): cells.append(cur.strip())
    return cells
def wrap(text, first, rest):
    return textwrap.fill(text, width=W, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False)
src=open(sys.argv[1]).read().split('\n')
out=[]; i=0; fence=False
while i<len(src):
    l=src[i]
    if l.startswith('|'):
        j=i
        while j<len(src) and src[j].startswith('|'): j+=1
        tbl=src[i:j]
        if max(len(x) for x in tbl)>W and len(tbl)>=2 and re.match(r'^\|[\s:|-]+\|?\s*$', tbl[1]):
            hdr=split_cells(tbl[0])
            for row in tbl[2:]:
                c=split_cells(row)
                if not c: continue
                out.append(wrap('* '+(c[0] if c[0] else '(—)'), '', '  '))
                for k in range(1,len(c)):
                    if c[k]=='' : continue
                    h=hdr[k] if k<len(hdr) and hdr[k] else f'col {k+1}'
                    out.append(wrap(f'- *{h}*: {c[k]}', '  ', '    '))
            out.append('')
        else:
            out.extend(tbl)
        i=j; continue
    if l.lstrip().startswith('```'):
        fence = not fence
    if len(l)>W and not fence and not l.lstrip().startswith('```'):
        lead=' '*(len(l)-len(l.lstrip()))
        m=re.match(r'^(\s*(?:[*-]|\d+\.)\s+)',l)
        out.append(wrap(l.strip(), lead, ' '*len(m.group(1)) if m else lead))
    else:
        out.append(l)
    i+=1
open(sys.argv[1],'w').write('\n'.join(out))
