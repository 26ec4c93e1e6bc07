# tools/profile.sh for every configuration the bench lines cover (probe sink of each scene, and the
# splat sink of the two scenes it is quoted on); results in gpurun_out/r03_<name>/.
set -e
for S in CornellBoxGlassMirrorDiffuse CornellBox MultiCameraTest CBGMD_stochastic128; do
  bash tools/profile.sh r03_$S $S > gpurun_out/r03_$S.log 2>&1
  echo "done $S"
done
for S in MirrorKnot4k MultiCameraTest; do
  bash tools/profile.sh r03_${S}_splat $S splat > gpurun_out/r03_${S}_splat.log 2>&1
  echo "done $S splat"
done
