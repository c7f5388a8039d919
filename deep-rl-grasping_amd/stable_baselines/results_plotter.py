from grasp_rl.sb.results_plotter import (X_EPISODES, X_TIMESTEPS, X_WALLTIME, LoadMonitorResultsError,  # noqa: F401
                                         get_monitor_files, load_results, ts2xy)
