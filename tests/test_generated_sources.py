def test_generated_apply_routine_is_what_its_generator_prints():
    """kiwi_amd/csrc/kiwi_apply_asm.inc (the apply of accumulate_multi_kernel as hand-allocated assembly) is generated text: the file in
    the tree is exactly what tools/gen_apply_asm.py prints, and no timing-experiment variant leaked into it."""
    import subprocess, sys, os
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kiwi_amd", "csrc")
    env = dict(os.environ)
    env.pop("KIWI_ASM_VARIANT", None)
    out = subprocess.run([sys.executable, os.path.join(csrc, "tools", "gen_apply_asm.py")], capture_output=True, text=True, env=env, check=True).stdout
    assert out == open(os.path.join(csrc, "kiwi_apply_asm.inc")).read()
    for routine in ("apply_group_asm_10_k5_rot_exact", "apply_group_asm_10_k9_plain_fused"):
        assert routine in out
    assert out.count("v_pk_fma_f32") > 500 and out.count("v_pk_add_f32") > 500
