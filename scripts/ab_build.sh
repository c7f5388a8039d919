#!/bin/bash
# development: same-box A/B of two BUILDS of the library.  Here (no GPU):   scripts/ab_build.sh build "-DGRL_RS_QUADS=1" engine
#   compiles the named translation units with the extra flags into build/obj_alt/ and links build/libgrl_alt.so (the other
#   units come from build/obj/).  On the GPU box:   scripts/ab_build.sh run [bench args]   alternates the in-tree library and
#   the alternative one (swapped in place in the box's scratch copy of the tree) and prints updates/s + per-launch times.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LIB=$R/deep-rl-grasping_amd/grasp_rl/libgrl.so
ALT=$R/build/libgrl_alt.so
if [ "$1" = build ]; then
  flags=$2; shift 2
  mkdir -p $R/build/obj_alt
  objs=""
  for u in engine gemm_fwd gemm_bwd gemm_wgrad heads; do
    if [[ " $* " == *" $u "* ]]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-kernarg-preload-count=16 $flags \
        -c $R/deep-rl-grasping_amd/csrc/$u.hip -o $R/build/obj_alt/$u.o || exit 1
      objs="$objs $R/build/obj_alt/$u.o"
    else
      objs="$objs $R/build/obj/$u.o"
    fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $ALT && echo "built $ALT ($flags: $*)"
  exit
fi
shift
q() { python $R/bench.py --no-learn-loop --no-success --no-cpu-baseline --repeats 3 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['step_kernel_ms']
print('%9.1f | ' % d['value'] + ' '.join('%s %.1f' % (n.replace('_fwd','F').replace('_bwd','B'), 1e3*v) for n, v in sorted(k.items(), key=lambda kv: -kv[1])))"; }
cp $LIB $R/build/libgrl_this.so
for i in 1 2; do
  cp $R/build/libgrl_this.so $LIB; echo -n "this  : "; q "$@"
  cp $ALT $LIB;                    echo -n "alt   : "; q "$@"
done
cp $R/build/libgrl_this.so $LIB
