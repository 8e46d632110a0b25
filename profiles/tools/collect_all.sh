set -o pipefail
bash profiles/tools/collect.sh default_r04 > gpurun_out/collect_default.log 2>&1 && echo default done
bash profiles/tools/collect.sh rgcn_r04 --model rgcn > gpurun_out/collect_rgcn.log 2>&1 && echo rgcn done
bash profiles/tools/collect.sh hgt_r04 --model hgt > gpurun_out/collect_hgt.log 2>&1 && echo hgt done
HET_SIDE_STREAM=0 HET_RGAT_OVERLAP=0 bash profiles/tools/collect.sh default_serial_r04 > gpurun_out/collect_default_serial.log 2>&1 && echo serial done
