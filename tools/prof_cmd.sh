#!/bin/bash
# rocprofv3 --kernel-trace --stats of one command (through gpurun): prints the top kernels.  usage: prof_cmd.sh <tag> <cmd...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
rm -rf /tmp/pc_$TAG
timeout ${PROF_TIMEOUT:-240} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc_$TAG -o x -- "$@" > /tmp/pc_$TAG.log 2>&1 < /dev/null
echo "rc=$?"
tail -${PROF_TAIL:-12} /tmp/pc_$TAG.log
f=$(find /tmp/pc_$TAG -name "*kernel_stats.csv" | head -1)
mkdir -p $R/gpurun_out
python3 -c "
import csv,sys,re
for i,row in enumerate(csv.reader(open(sys.argv[1]))):
    n=row[0]; n=re.sub(r'\(anonymous namespace\)::','',n); n=re.sub(r'^void ','',n); n=n.split('(')[0] if not n.startswith('void at') else n[:60]
    print('%-70s %s' % (n[:70], ' '.join(row[1:4]+row[5:7])))
" "$f" | head -${PROF_ROWS:-25} | tee $R/gpurun_out/${TAG}_kernel_stats.txt
