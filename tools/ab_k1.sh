#!/bin/bash
# A/B of K1 build configurations.  Local: `tools/ab_k1.sh build name "<extra hipcc flags>"` cross-compiles
# csrc/variants/libbdf_<name>.so; on the GPU box: `tools/ab_k1.sh run [steps]` benches every variant.
set -e
root=$(cd $(dirname $0)/.. && pwd)
src=$root/bayesiandatafusion.jl_amd/csrc
var=$src/variants
mkdir -p $var
if [ "$1" = build ]; then
  name=$2; flags=$3
  tmp=$(mktemp -d)
  for f in $src/*.hip; do
    o=$tmp/$(basename ${f%.hip}).o
    extra=""
    case "$(basename $f)" in
      k_sample_rows.hip|k_rows_col.hip) extra="$flags" ;;
      bdf_api.hip) case "$flags" in *BDF_K1_STAMPS*) extra="-DBDF_K1_STAMPS" ;; *BDF_K1_SPANS*) extra="-DBDF_K1_SPANS" ;; esac ;;
      k_hyper.hip) case "$flags" in *BDF_HYPER_STAMPS*) extra="-DBDF_HYPER_STAMPS" ;; esac ;;
      k_feat.hip) case "$flags" in *BDF_CG_STAMPS*) extra="-DBDF_CG_STAMPS" ;; *BDF_CG_RM*) extra="$flags" ;; esac ;;
      k_predict.hip) case "$flags" in *BDF_PREDICT*) extra="$flags" ;; esac ;;
    esac
    if [ -z "$extra" ] && [ -f ${f%.hip}.o ]; then cp ${f%.hip}.o $o; continue; fi
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 $extra -c $f -o $o
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $var/libbdf_$name.so $tmp/*.o
  rm -rf $tmp
  echo built $var/libbdf_$name.so
else
  steps=${2:-200}
  for so in $var/libbdf_*.so; do
    echo "== $(basename $so)"
    BDF_LIB_PATH=$so python3 $root/bench.py --steps $steps --warmup 20 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['roofline'])"
  done
fi
