# same-box A/B of the LDS-tile depthwise kernel's rows per thread (variant libraries built with -DDS_DW_SR=..) on the bf16x3 tier's layer shapes
for lib in "$@"; do echo $lib; DS_LIB=$lib bash tools/dw_fp32_layers.sh; done
