#!/usr/bin/env bash
# tools/memguard.sh [-m GiB] [-t seconds] cmd args... -- runs cmd as a child and ends it (TERM, then KILL) when its resident set
# passes the limit (default 32 GiB) or the time limit (default 900 s) is up.  A GPU box that runs out of host memory is lost for
# everyone on it: every bench / test / probe command of tools/*.sh goes through this guard.
LIM_GIB=32; LIM_S=900
while getopts "m:t:" o; do case $o in m) LIM_GIB=$OPTARG;; t) LIM_S=$OPTARG;; esac; done
shift $((OPTIND - 1))
"$@" &
pid=$!
start=$(date +%s)
while kill -0 $pid 2>/dev/null; do
  rss_kb=0
  for p in $pid $(pgrep -P $pid 2>/dev/null); do
    r=$(awk '/^VmRSS:/{print $2}' /proc/$p/status 2>/dev/null); rss_kb=$((rss_kb + ${r:-0}))
  done
  now=$(date +%s)
  if [ $rss_kb -gt $((LIM_GIB * 1048576)) ] || [ $((now - start)) -gt $LIM_S ]; then
    echo "memguard: ending $pid (rss ${rss_kb} kB, $((now - start)) s)" >&2
    kill -TERM $pid 2>/dev/null; sleep 2; kill -KILL $pid 2>/dev/null; pkill -KILL -P $pid 2>/dev/null
    wait $pid 2>/dev/null; exit 137
  fi
  sleep 0.2
done
wait $pid
