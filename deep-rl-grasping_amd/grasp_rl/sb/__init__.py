"""The part of the stable-baselines 2.10 object API that BarisYazici/deep-rl-grasping calls
(SURVEY.md 8b), re-implemented over the HIP engine.  ``deep-rl-grasping_amd/stable_baselines`` re-exports
these modules under the import paths the reference's scripts use."""
