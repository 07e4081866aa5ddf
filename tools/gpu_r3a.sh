YGZF_DESC_PIPE=1 YGZF_LIBRARY=$PWD/orb_ygz_slam_amd/lib_ab/libygzf_pclk.so python tools/desc_clk_probe.py
