for spec in "target_veh3dof_fhadp_b4096_h30 fp32" "cfg2_idp_fhadp_b4096_h30 fp32" "cfg3_veh3dof_infadp_b8192 fp32" "cfg4_veh3dof_fhadp_b4096_h50 fp32" "cfg5_lq_infadp_b65536 fp32" "cfg5_lq_infadp_b65536 fp16" "cfg1_idp_fhadp_b64_h10 fp32"; do
  set -- $spec
  bash tools/collect_profile.sh ${TAG:-r04} $1 $2 2>&1 | grep -E "pmc pass|rror" | tr '\n' ' '; echo " <- $1 $2"
done
