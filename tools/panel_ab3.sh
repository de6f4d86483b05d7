cd $GRAFT_REPO_ROOT
timeout -k 5 60 build/panel_probe || exit 1
bash tools/panel_ab2.sh
