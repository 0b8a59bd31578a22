"""FABind+ (KDD'25 model, reference FABind_plus/fabind/) on the MI355X kernels -- SURVEY.md row a18.

`fabind_amd.plus.models` mirrors `FABind_plus/fabind/models/*` (module / parameter names, constructor signatures);
`fabind_amd.plus.engine` drives the HIP kernels.  Round 1 ships the layer stack (`EfficientMCAttModel`: forward,
fp32 parity + bf16), see DESIGN.md section 8."""
