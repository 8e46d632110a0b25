# every committed profile of a round (run on the GPU box through gpurun, from the repo root): bash profiles/tools/collect_all.sh
set -o pipefail
bash profiles/tools/collect.sh default_r05 > gpurun_out/collect_default.log 2>&1 && echo default done
HET_SIDE_STREAM=0 HET_RGAT_OVERLAP=0 bash profiles/tools/collect.sh default_serial_r05 > gpurun_out/collect_default_serial.log 2>&1 && echo serial done
if [ "$1" != "rgat" ]; then
bash profiles/tools/collect.sh rgcn_r05 --model rgcn > gpurun_out/collect_rgcn.log 2>&1 && echo rgcn done
bash profiles/tools/collect.sh hgt_r05 --model hgt > gpurun_out/collect_hgt.log 2>&1 && echo hgt done
fi
