#!/bin/bash
# LLVM scheduler-option grid for the two forward+adjoint translation units of K3 (the optimum moves with the source:
# rounds 1-2 found 3-7 % between option sets at an identical instruction mix; csrc/Makefile ships the winner).
#   bash tools/sched_grid.sh build      here: one library per option set -> tools/_build/libsvbrdf_g<set>.so
#   bash tools/sched_grid.sh run        on the GPU box: same-box A/B of all of them (tools/k3_split_bench)
cd "$(dirname "$0")/.."
NP="-mllvm -enable-post-misched=0"
MMC="-mllvm -amdgpu-sched-strategy=max-memory-clause"
R4="-mllvm -greedy-regclass-priority-trumps-globalness=1 -mllvm -greedy-reverse-local-assignment"
declare -A SETS
SETS[X]="$NP -mllvm -amdgpu-sched-strategy=iterative-minreg $R4"      # shipped since round 2
SETS[Y]="$NP -mllvm -amdgpu-sched-strategy=iterative-maxocc $R4"
SETS[T]="$NP -mllvm -amdgpu-sched-strategy=iterative-minreg"
SETS[S]="$NP -mllvm -amdgpu-sched-strategy=iterative-maxocc"
SETS[Z]="$NP -mllvm -amdgpu-sched-strategy=iterative-ilp"
SETS[A]="$NP $MMC"
SETS[G]="$NP $MMC $R4"
SETS[H]="-mllvm -misched-postra-direction=bottomup $MMC"
SETS[M]="-mllvm -misched-postra-direction=bottomup $MMC $R4"
SETS[E]="$NP -mllvm -amdgpu-sched-strategy=max-ilp"
SETS[B]="-mllvm -misched-postra-direction=bottomup"
SETS[D]="$NP"
SETS[V]="-mllvm -amdgpu-sched-strategy=iterative-minreg $R4"           # X with the post-RA scheduler left on
SETS[W]="-mllvm -misched-postra-direction=bottomup -mllvm -amdgpu-sched-strategy=iterative-minreg $R4"
if [ "$1" = build ]; then
  for k in "${!SETS[@]}"; do SCHED_ADJOINT="${SETS[$k]}" bash tools/build_variant.sh g$k "${@:2}" 2>&1 | tail -1; done
  for k in "${!SETS[@]}"; do echo "SETS[$k]=\"${SETS[$k]}\""; done | sort > tools/_build/sched_grid_sets.txt
else
  cat tools/_build/sched_grid_sets.txt
  B=tools/_build
  for round in 1 2; do echo "== round $round"
    for cfg in "tied:" "untied:K3_UNTIED=1" "mixed:K3_L1=0.1" "head+l1:K3_HEAD=1,K3_L1=0.1"; do
      tag=${cfg%%:*}; envs=${cfg#*:}
      for lib in $B/libsvbrdf_r4.so $(ls $B/libsvbrdf_g?.so | sort); do
        printf "%-8s %-4s " "$tag" "$(basename $lib .so | sed s/libsvbrdf_//)"
        env ${envs//,/ } K3_LIB=$lib K3_MODES=04 K3_ROUNDS=1 K3_STEPS=600 $B/k3_split_bench | awk '{printf "%s us  ", $(NF-5)} END {print ""}'
      done
    done
  done
fi
