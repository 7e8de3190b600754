export LD_LIBRARY_PATH=$PWD/autobub3hs_amd:$LD_LIBRARY_PATH
M=./tools/k2_microbench
( echo "# k3 sigma2 kf2 / kf1 (discs)"; $M 2000 5 0 1280 1024 0 2 1 0 1 1; ABUB_K3_KF=1 $M 2000 5 0 1280 1024 0 2 1 0 1 1;
  echo "# k3 sigma2 kf2 / kf1 (nodisc)"; $M 2000 5 0 1280 1024 0 2 1 0 0 1; ABUB_K3_KF=1 $M 2000 5 0 1280 1024 0 2 1 0 0 1;
  echo "# k3 sigma2 cycle8 kf2 / kf1 (L2 resident)"; $M 2000 5 0 1280 1024 0 2 1 8 0 1; ABUB_K3_KF=1 $M 2000 5 0 1280 1024 0 2 1 8 0 1;
  echo "# k3 sigma2 kf2 / kf1 1680"; $M 2000 5 0 1680 1050 0 2 1 0 1 1; ABUB_K3_KF=1 $M 2000 5 0 1680 1050 0 2 1 0 1 1 ) 2>&1
