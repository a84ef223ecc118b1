#!/usr/bin/env python3
"""Reflow a Markdown file to <= 118 bytes per line: paragraphs and list items are re-joined and wrapped (continuation
lines keep the item's indent), tables whose rows do not fit become bullet lists ("- cell | cell | ..."), fenced code,
headings and short tables stay.  python tools/reflow_md.py IN OUT"""
import re, sys, textwrap
LIMIT = 118
ITEM = re.compile(r'^(\s*)([-*+]|\d+\.)\s+')
def wrap(text, first, rest):
  w = 116
  while True:
    lines = textwrap.wrap(text, width=w, initial_indent=first, subsequent_indent=rest, break_long_words=False,
                          break_on_hyphens=False) or [first.rstrip()]
    if all(len(l.encode()) <= LIMIT for l in lines) or w < 60:
      return lines
    w -= 4
src = open(sys.argv[1]).read().split('\n')
out, i = [], 0
def special(l):
  s = l.lstrip()
  return (not s) or s.startswith(('#', '|', '```')) or bool(ITEM.match(l))
while i < len(src):
  line = src[i]
  s = line.lstrip()
  if s.startswith('```'):
    j = i + 1
    while j < len(src) and not src[j].lstrip().startswith('```'):
      j += 1
    out.extend(src[i:j + 1]); i = j + 1; continue
  if not s or s.startswith('#'):
    out.append(line); i += 1; continue
  if s.startswith('|'):
    j = i
    while j < len(src) and src[j].lstrip().startswith('|'):
      j += 1
    rows = src[i:j]
    if all(len(r.encode()) <= LIMIT for r in rows):
      out.extend(rows)
    else:
      header = None
      for r in rows:
        cells = [c.strip() for c in r.strip().strip('|').split('|')]
        if all(re.fullmatch(r':?-+:?', c or '-') for c in cells):
          continue
        if header is None:
          header = cells
          out.extend(wrap('(' + ' | '.join(cells) + ')', '', '  '))
          continue
        out.extend(wrap(' | '.join(cells), '- ', '  '))
      out.append('')
    i = j; continue
  m = ITEM.match(line)
  first = m.group(0) if m else re.match(r'^\s*', line).group(0)
  rest = ' ' * len(first) if m else first
  parts = [line[len(first):].strip()]
  j = i + 1
  while j < len(src) and not special(src[j]):
    parts.append(src[j].strip()); j += 1
  out.extend(wrap(' '.join(parts), first, rest))
  i = j
open(sys.argv[2], 'w').write('\n'.join(out))
