#!/bin/bash
# the round's evidence of the final tree (profiles/r03 via tools/publish_profiles_r03.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; bash tools/collect_profiles_r03.sh
