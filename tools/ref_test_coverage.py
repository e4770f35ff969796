"""Which of the reference's own tests for SURVEY 8 f3 are restated in tests/?  (build container only: reads /root/reference.)
For every #[test] / def test_ of the reference's rule, env, observation and mapper modules: is there a citation of a line inside
that test's span in tests/test_shogi*.py, tests/test_hip_shogi_env.py or the oracle?  `explicit` = written as file.rs:LINE;
`any` = additionally the abbreviated form (":LINE-LINE" behind a file named earlier in the same docstring).  Tests that exercise
data structures this build does not have are listed in NOT_APPLICABLE with the reason.   python tools/ref_test_coverage.py"""
import glob, os, re, sys
REF = "/root/reference/shogi-engine/crates/"
FILES = ["shogi-core/src/rules.rs", "shogi-core/src/game.rs", "shogi-core/src/movegen.rs", "shogi-core/src/attack.rs", "shogi-gym/src/vec_env.rs",
         "shogi-gym/src/katago_observation.rs", "shogi-gym/src/observation.rs", "shogi-gym/src/spatial_action_mapper.rs",
         "shogi-gym/src/action_mapper.rs", "shogi-gym/src/step_result.rs", "shogi-gym/tests/test_vec_env.py",
         "shogi-gym/tests/test_observation.py", "shogi-gym/tests/test_action_mapper.py"]
NOT_APPLICABLE = {
    "incremental attack map / ray updates / would_wrap_file helper (the oracle and the device kernel recompute attacks per query)":
        r"attack\.rs::test_(incremental|update_rays|would_wrap)",
    "make / unmake and incremental hash / pawn-column state (nothing is unmade here: legality is decided on an overlay)":
        r"game\.rs::test_(make_unmake|hash_matches|attack_map_matches|unmake|deep_make_unmake|hot_path|multi_ply_hash|pawn_columns_after|compute_pawn_columns|full_game_make_unmake|from_position)",
    "SFEN parsing, spectator dictionaries (web UI feeds: DESIGN section 7)":
        r"(game\.rs::test_from_sfen|test_vec_env\.py::test_get_spectator|test_vec_env\.py::test_get_sfen_matches_spectator)",
    "Rust panic isolation / caller-provided buffer length checks (no such API surface: the env owns its buffers)":
        r"(vec_env\.rs::test_(apply_moves|katago_spatial_apply_moves)|katago_observation\.rs::test_(wrong_buffer_length|katago_observation_wrong_buffer_length|from_position_inserts))",
}
def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ours = "".join(open(f).read() for f in sorted(glob.glob(root + "/tests/test_shogi*.py")) + [root + "/tests/test_hip_shogi_env.py", root + "/oracle/shogi.py", root + "/oracle/shogi_oracle.c"])
    explicit, loose = {}, set()
    for m in re.finditer(r"([a-z_]+\.(?:rs|py)):(\d+)(?:-(\d+))?", ours):
        explicit.setdefault(m.group(1), []).append((int(m.group(2)), int(m.group(3) or m.group(2))))
    for m in re.finditer(r"(?<![\w.]):(\d{2,4})(?:-(\d{2,4}))?|,\s?(\d{2,4})-(\d{2,4})", ours):
        loose.add(int(m.group(1) or m.group(3)))
    rows, tot = [], [0, 0, 0, 0]
    for f in FILES:
        lines, base = open(REF + f).read().split("\n"), os.path.basename(f)
        tests = []
        if f.endswith(".rs"):
            starts = [i for i, l in enumerate(lines) if "#[test]" in l]
        else:
            starts = [i for i, l in enumerate(lines) if re.match(r"\s*def test_", l)]
        for n, i in enumerate(starts):
            j = i
            while not re.search(r"(fn|def) (\w+)", lines[j]): j += 1
            tests.append((re.search(r"(fn|def) (\w+)", lines[j]).group(2), i + 1, starts[n + 1] if n + 1 < len(starts) else len(lines)))
        ex = lo = na = 0
        missing = []
        for name, a, b in tests:
            key = f"{base}::{name}"
            if any(re.search(pat, key) for pat in NOT_APPLICABLE.values()):
                na += 1
            elif any(l <= b and h >= a for l, h in explicit.get(base, [])):
                ex += 1
            elif any(a <= v <= b for v in loose):
                lo += 1
            else:
                missing.append(f"{name}:{a}")
        rows.append((f, len(tests), ex, lo, na, missing))
        for k, v in enumerate((len(tests), ex, lo, na)): tot[k] += v
    print(f"{'reference file':46s} tests  cited(file:line)  cited(:line only)  n/a  uncited")
    for f, n, ex, lo, na, missing in rows:
        print(f"{f:46s} {n:5d} {ex:17d} {lo:18d} {na:4d}  {len(missing)}" + ("  " + ", ".join(missing) if missing else ""))
    print(f"{'total':46s} {tot[0]:5d} {tot[1]:17d} {tot[2]:18d} {tot[3]:4d}  {tot[0] - tot[1] - tot[2] - tot[3]}")
    for why, pat in NOT_APPLICABLE.items():
        print(f"n/a: {why}")
if __name__ == "__main__":
    main()
