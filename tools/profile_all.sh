#!/bin/bash
# The round's whole measurement set on the GPU box: tools/profile_round.sh for every bench configuration, then the default
# bench line (headline + companion).  Everything lands under gpurun_out/<tag>/; copy what is to be kept into profiles/.
# usage: tools/profile_all.sh <tag e.g. r03_d>
TAG=${1:?tag}
cd "$GRAFT_REPO_ROOT"
bash tools/profile_round.sh $TAG c3_sponza1080p "sponza_lod 1920x1080 1spp 5-bounce" --no-companion --no-own-tree
bash tools/profile_round.sh $TAG atrium1080p "atrium 1920x1080 1spp 5-bounce" --scene atrium
bash tools/profile_round.sh $TAG c2_cornell1080p "cornell 1920x1080 1spp 5-bounce" --config c2
bash tools/profile_round.sh $TAG c5_sponza1080p_svgf "sponza_lod 1920x1080 1spp 5-bounce svgf" --config c5
bash tools/profile_round.sh $TAG c4_atrium4k8spp "atrium 3840x2160 8spp 8-bounce" --config c4
timeout 900 python bench.py > gpurun_out/$TAG/${TAG}_bench_default.json 2> gpurun_out/$TAG/default.err
tail -c 600 gpurun_out/$TAG/${TAG}_bench_default.json
