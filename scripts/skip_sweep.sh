run() { echo "SKIP=$1"; STEGAN_SKIP=$1 python scripts/whatif2.py 2>&1 | grep WHATIF | sed 's/WHATIF= WHATIF_C= //'; }
run ""
run stegan_conv1d_bwd_weight_hx
run stegan_conv1d_bwd_weight_ws
run stegan_conv1d_bwd_weight_skinny,stegan_conv1d_bwd_weight_skinny_multi
run stegan_conv1d_fwd_skinny,stegan_conv1d_fwd_skinny_multi,stegan_conv1d_bwd_data_skinny,stegan_conv1d_bwd_data_skinny_multi
run stegan_add_layernorm_fwd_img,stegan_add_layernorm_bwd_img
run stegan_relpos_attention_fwd_img,stegan_relpos_attention_bwd_img
run stegan_spectral_norm_fwd_multi,stegan_spectral_norm_bwd_acc
run stegan_act_bwd_add_img,stegan_act_bwd_add_multi_img
run stegan_wn_backward_z
run stegan_pair_loss_multi,stegan_pair_loss_multi_ex
run stegan_multi_td_fwd_bwd_ex
run stegan_fold_upsample2_img
