# bench.py of an older tree (a git worktree under the repo root, built in place) against this tree, alternating on one box:
# box-to-box spread (+-0.1 ms) is larger than a round's gain.  tools/compare_trees.sh _r04_tree "round-4 tree"
OLD=$1; NAME=${2:-older tree}
line() { python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], 'ms per step', d['value'], 'samples/s')"; }
for i in 1 2 3; do
  (cd $OLD && python bench.py --no-variants --no-cpu-baseline 2>/dev/null | line "$NAME")
  python bench.py --no-variants --no-cpu-baseline 2>/dev/null | line "this tree"
done
(cd $OLD && python bench.py --no-variants --no-cpu-baseline --surface 2>/dev/null | line "$NAME, surface scenes")
python bench.py --no-variants --no-cpu-baseline --surface 2>/dev/null | line "this tree, surface scenes"
