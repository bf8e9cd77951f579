/* placeholder, filled in later */
